image_size = 256
load_vae_feat = False
vae_pretrained = "does/not/exist"
train_batch_size = 2
