# BASELINE config 5: Segformer-B1 student <- B4 teacher, CGD on the 4 decoder features (token-major taps linear_c1..4,
# E 256 -> 768 through 1x1 align convs), no resize (spatial dims already equal), bf16 activations; bs 64 = 8 per GPU x 8 GPUs.
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

_stage = lambda i: dict(student_layer=f'decode_head.linear_c{i}', teacher_layer=f'decode_head.linear_c{i}', loss_name='KLDLoss',  # noqa: E731
                        channel_nums=(256, 768),
                        loss_config=dict(alpha=3, tau=4, shuffle_config={'interval': 1000},
                                         transform_config={'loss_type': 'channel', 'group_size': 8}))
model = dict(
    type='SDModule',
    cfg_s=_segformer('b1', [64, 128, 320, 512], 256),
    cfg_t=_segformer('b4', [64, 128, 320, 512], 768),
    distillation=[_stage(1), _stage(2), _stage(3), _stage(4)],
    t_pretrain='./pretrained/segformer.b4.512x512.ade.160k.pth',
    train_cfg=dict(),
    test_cfg=dict(mode='whole'))
data = dict(samples_per_gpu=8)
precision = dict(activations='bf16')  # bf16 storage of the tapped features / autocast of the networks, fp32 accumulation
