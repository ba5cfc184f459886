"""Hyper-parameters of the one-shot fine-tune head (NOT the clustering model); values follow the
reference's configs/trainers/supervised_config.py."""
lr, beta1, beta2, weight_decay = 0.001, 0.9, 0.99, 0.0
losses, lambdas = ['cross_entropy'], [1.]
scheduler_type, scheduler_args = 'step', dict(step_size=500, gamma=0.1)
num_epochs, print_freq = 200, 10
