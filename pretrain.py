"""Pre-train the self-supervised per-pixel clustering model (MI355X-native hot path).

Same flags, choices and method remap as the reference's pretrain.py (:23-93).  Opt-in extras:
--synthetic (random-init generator + random latents: no checkpoints exist offline),
--batch / --sinkhorn_scope / --epochs (override the config), multi-GPU via torch.distributed.run.
"""
import argparse
import os

MODELS = ['ffhq-256', 'cat-256', 'afhq-256', 'horse-256', 'car-512', 'pidray-256', 'pidray-pliers-256',
          'pidray-hammer-256', 'pidray-powerbank-256', 'pidray-wrench-256', 'pidray-handcuffs-256']
SWAV_BY_MODEL = (('ffhq-256', 'ffhq'), ('cat-256', 'cat'), ('car-512', 'car'), ('horse-256', 'horse'))


def remap(method, model):
    """hfc_with_swav -> hfc_with_swav_<dataset> (pretrain.py:72-81)"""
    if method != 'hfc_with_swav':
        return method
    for m, tag in SWAV_BY_MODEL:
        if model == m:
            return 'hfc_with_swav_' + tag
    return 'hfc_with_swav_pidray' if 'pidray' in model else method


def main():
    p = argparse.ArgumentParser(description="Pre-train the self-supervised clustering model for one-shot segmentation")
    p.add_argument("--model", default='ffhq-256', choices=MODELS, type=str)
    p.add_argument("--method", default='hfc_with_swav', choices=['hfc_with_swav', 'hfc_with_simclr', 'hfc_kmeans'], type=str)
    p.add_argument("--out_dir", default="results/pretrain_default_ffhq/")
    p.add_argument("--expt_desc", default="Testing Clustering Model")
    p.add_argument("--num_test_samples", default=10, type=int)
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--batch", type=int, default=None)
    p.add_argument("--sinkhorn_scope", default=None, choices=['image', 'global'])
    p.add_argument("--epochs", type=int, default=None)
    args = p.parse_args()
    args.method = remap(args.method, args.model)

    import torch
    group = None
    device = 'cuda:0'
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        device = f'cuda:{local}'
        dist.init_process_group("nccl", device_id=torch.device(device))
        group = dist.group.WORLD

    from src.one_shot_pipeline import OneShotPipeline
    pipe = OneShotPipeline(out_dir=args.out_dir, exp_name=args.expt_desc, model=args.model, segmentor=args.method,
                           num_test_samples=args.num_test_samples, device=device, synthetic=args.synthetic,
                           dist_group=group)
    pipe.seg_config.train_hfc = True
    pipe.seg_config.hfc_prep_args['train'] = True
    if args.method == 'hfc_kmeans':
        pipe.seg_config.hfc_prep_args['presaved'] = False
    prep = pipe.seg_config.hfc_prep_args
    if 'swav_args' in prep:
        if args.batch:
            prep['swav_args']['batch_size'] = args.batch
        if args.sinkhorn_scope:
            prep['swav_args']['sinkhorn_scope'] = args.sinkhorn_scope
        if args.epochs:
            prep['swav_args']['num_epochs'] = args.epochs
    pipe.run_pipeline()


if __name__ == "__main__":
    main()
