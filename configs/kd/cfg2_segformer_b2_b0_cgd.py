# BASELINE config 2 (headline): Segformer-B0 student <- B2 teacher, CGD group=8 T=4 alpha=3, bs 8 per GPU, 512 x 512
# = reference local_configs/Teacher_Student_Size/b2b0.py + loss_config of Weight_Temperature/w=3_t=4.py / Group_Size/cgd10.py
_base_ = ['../_base_/synthetic_ade20k.py', '../_base_/default_runtime.py', '../_base_/schedule_160k_adamw.py']
norm_cfg = dict(type='SyncBN', requires_grad=True)


def _segformer(variant, in_channels, embed_dim):
    # field values as in reference local_configs/Teacher_Student_Size/b2b0.py:8-106
    return dict(
        type='EncoderDecoder',
        pretrained=f'pretrained/mit_{variant}.pth',
        backbone=dict(type=f'mit_{variant}', style='pytorch'),
        decode_head=dict(type='SegFormerHead', in_channels=in_channels, in_index=[0, 1, 2, 3], feature_strides=[4, 8, 16, 32],
                         channels=128, dropout_ratio=0.1, num_classes=150, norm_cfg=norm_cfg, align_corners=False,
                         decoder_params=dict(embed_dim=embed_dim),
                         loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)))

model = dict(
    type='SDModule',
    cfg_s=_segformer('b0', [32, 64, 160, 256], 256),
    cfg_t=_segformer('b2', [64, 128, 320, 512], 768),
    distillation=[dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.linear_pred', loss_name='CGDLoss',
                       loss_config={'group_size': 8, 'alpha': 3, 'tau': 4})],
    t_pretrain='./pretrained/segformer.b2.512x512.ade.160k.pth',
    train_cfg=dict(),
    test_cfg=dict(mode='whole'))
data = dict(samples_per_gpu=8)
