# BASELINE config 1: PSPNet-R18 student <- frozen PSPNet-R101 teacher, channel-wise logit KL (CDLoss), 2 x 3 x 512 x 512
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

model = dict(
    type='SDModule',
    cfg_s=_pspnet(18, 512, 256, 128, 64),
    cfg_t=_pspnet(101, 2048, 1024, 512, 256),
    distillation=[dict(student_layer='decode_head.conv_seg', teacher_layer='decode_head.conv_seg', loss_name='CDLoss', loss_config={})],
    t_pretrain='./pretrained/pspnet_r101-d8_512x512_160k_ade20k.pth',
    train_cfg=dict(),
    test_cfg=dict(mode='whole'))
data = dict(samples_per_gpu=2)
