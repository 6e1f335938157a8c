# BASELINE config 4: PSPNet-R18 student <- Swin-B UPerNet teacher, CGD on decoder features through a trainable 1x1
# align conv 128 -> 512 (MFMA kernel); student decode_head.bottleneck [B,128,64,64] vs teacher decode_head.fpn_bottleneck
# [B,512,128,128]; both resized to the label size is wasteful, so the softmax runs at the teacher's 128x128
# (resize_config target='teacher': the aligned student feature is resized x2 and the criterion reads both at 128x128);
# group_size 16 (512 % 16 == 0, no padding).  bs 32 = 8 per GPU x 4 GPUs.
_base_ = ['../_base_/synthetic_ade20k.py', '../_base_/default_runtime.py', '../_base_/schedule_160k_adamw.py']
norm_cfg = dict(type='SyncBN', requires_grad=True)


def _pspnet(depth, c4, c3, head_ch, aux_ch):
    # reference configs/_base_/models/pspnet_r50-d8.py with num_classes=150; R18 dims from
    # configs/pspnet/pspnet_r18-d8_512x1024_80k_cityscapes.py, R101 from pspnet_r101-d8_512x512_80k_ade20k.py
    return dict(
        type='EncoderDecoder',
        pretrained=None,
        backbone=dict(type='ResNetV1c', depth=depth, num_stages=4, out_indices=(0, 1, 2, 3), dilations=(1, 1, 2, 4),
                      strides=(1, 2, 1, 1), norm_cfg=norm_cfg, norm_eval=False, style='pytorch', contract_dilation=True),
        decode_head=dict(type='PSPHead', in_channels=c4, in_index=3, channels=head_ch, pool_scales=(1, 2, 3, 6), dropout_ratio=0.1,
                         num_classes=150, norm_cfg=norm_cfg, align_corners=False,
                         loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
        auxiliary_head=dict(type='FCNHead', in_channels=c3, in_index=2, channels=aux_ch, num_convs=1, concat_input=False,
                            dropout_ratio=0.1, num_classes=150, norm_cfg=norm_cfg, align_corners=False,
                            loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)))


# Swin-B UPerNet teacher; layout of reference local_configs/_base_/models/upernet_swin.py, 150 classes
upernet_swin_b = dict(
    type='EncoderDecoder',
    pretrained=None,
    backbone=dict(type='SwinTransformer', embed_dim=128, depths=[2, 2, 18, 2], num_heads=[4, 8, 16, 32], window_size=7, mlp_ratio=4.,
                  qkv_bias=True, qk_scale=None, drop_rate=0., attn_drop_rate=0., drop_path_rate=0.3, ape=False, patch_norm=True,
                  out_indices=(0, 1, 2, 3), use_checkpoint=False),
    decode_head=dict(type='UPerHead', in_channels=[128, 256, 512, 1024], in_index=[0, 1, 2, 3], pool_scales=(1, 2, 3, 6), channels=512,
                     dropout_ratio=0.1, num_classes=150, norm_cfg=norm_cfg, align_corners=False,
                     loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0)),
    auxiliary_head=dict(type='FCNHead', in_channels=512, in_index=2, channels=256, num_convs=1, concat_input=False, dropout_ratio=0.1,
                        num_classes=150, norm_cfg=norm_cfg, align_corners=False,
                        loss_decode=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=0.4)))
model = dict(
    type='SDModule',
    cfg_s=_pspnet(18, 512, 256, 128, 64),
    cfg_t=upernet_swin_b,
    distillation=[dict(student_layer='decode_head.bottleneck', teacher_layer='decode_head.fpn_bottleneck', loss_name='KLDLoss',
                       channel_nums=(128, 512),
                       loss_config=dict(alpha=3, tau=4, resize_config=dict(mode='bilinear', align_corners=False, target='teacher'),
                                        shuffle_config={'interval': 1000},
                                        transform_config={'loss_type': 'channel', 'group_size': 16}))],
    t_pretrain='./pretrained/upernet_swin_base_patch4_window7_512x512.pth',
    train_cfg=dict(),
    test_cfg=dict(mode='whole'))
data = dict(samples_per_gpu=8)
