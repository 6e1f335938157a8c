# runtime defaults in the reference's config dialect (cf. reference local_configs/_base_/default_runtime.py)
log_config = dict(interval=50, hooks=[dict(type='TextLoggerHook', by_epoch=False)])
dist_params = dict(backend='nccl')  # 'nccl' IS RCCL on ROCm
log_level = 'INFO'
load_from = None
resume_from = None
workflow = [('train', 1)]
cudnn_benchmark = True
