# synthetic ADE20K-shaped data: 512x512 crops, 150 classes, ignore index 255
dataset_type = 'SyntheticADE'
crop_size = (512, 512)
num_classes = 150
data = dict(samples_per_gpu=2, workers_per_gpu=0)
