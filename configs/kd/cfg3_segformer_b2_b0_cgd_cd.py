# BASELINE config 3: config 2 + channel-wise logit KL on the same taps; bs 64 = 8 per GPU x 8 GPUs.
# Two KLDLoss entries with distinct transform_config (two preset losses on one layer pair collide in the reference).
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

_resize = dict(mode='bilinear', align_corners=False)
model = dict(
    type='SDModule',
    cfg_s=_segformer('b0', [32, 64, 160, 256], 256),
    cfg_t=_segformer('b2', [64, 128, 320, 512], 768),
    distillation=[
        dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.linear_pred', loss_name='KLDLoss',
             loss_config=dict(alpha=3, tau=4, resize_config=_resize, shuffle_config={'interval': 1000},
                              transform_config={'loss_type': 'channel', 'group_size': 8})),
        dict(student_layer='decode_head.linear_pred', teacher_layer='decode_head.linear_pred', loss_name='KLDLoss',
             loss_config=dict(alpha=1, tau=1, resize_config=_resize,
                              transform_config={'loss_type': 'channel', 'group_size': 1})),
    ],
    t_pretrain='./pretrained/segformer.b2.512x512.ade.160k.pth',
    train_cfg=dict(),
    test_cfg=dict(mode='whole'))
data = dict(samples_per_gpu=8)
