"""Entry point — counterpart of reference train.py:74-204 for the models on the accelerated path
(MF, NGCF, CDAE).  hydra / wandb are not required: the config is ``configs/train_config.yaml`` in
the reference's layout (or the built-in defaults) plus ``key=value`` overrides.

    python -m yelprecommendation_amd.train model_name=MF data_dir=data/ epochs=5 batch_size=4096
    python -m yelprecommendation_amd.train model_name=MF synthetic=yelp2018 fast_loader=true

Call order is the reference's: pipeline.preprocess() -> split() -> datasets -> set_seed() ->
DataLoaders -> trainer.run() -> load_best_model() -> evaluate(test).
"""
import logging
import sys

import pandas as pd
from torch.utils.data import DataLoader

from .utils import Config, load_config, logger, make_config, set_seed


def _parse_overrides(argv):
    out = {}
    for a in argv:
        if "=" not in a:
            raise SystemExit(f"expected key=value, got {a!r}")
        k, v = a.split("=", 1)
        for cast in (int, float):
            try:
                v = cast(v)
                break
            except ValueError:
                continue
        else:
            if v.lower() in ("true", "false"):
                v = v.lower() == "true"
        out[k] = v
    return out


def build(cfg):
    """reference train.py:134-192: pipeline, split, datasets, model_info."""
    from .data.datasets.mf_data_pipeline import MFDataPipeline
    from .data.datasets.mf_dataset import MFDataset
    from .data.datasets.ngcf_data_pipeline import NGCFDataPipeline
    args = Config()
    if cfg.model_name == 'MF':
        pipe = MFDataPipeline(cfg)
    elif cfg.model_name == 'NGCF':
        pipe = NGCFDataPipeline(cfg)
    elif cfg.model_name == 'CDAE':
        from .data.datasets.cdae_data_pipeline import CDAEDataPipeline
        pipe = CDAEDataPipeline(cfg)
    else:
        raise ValueError(f"model '{cfg.model_name}' is not on the accelerated path (MF, NGCF, CDAE)")
    synthetic = cfg.get("synthetic")
    if synthetic:
        from .data.synthetic import make_frame
        if synthetic == "yelp2018":
            import torch
            if str(cfg.device).lower() == "cuda" and torch.cuda.is_available():
                # same generative model drawn on the GPU (seconds instead of ~30 s of NumPy)
                import pandas as pd
                from .data.synthetic import make_interactions_torch
                u, i = make_interactions_torch(mean_items=47.0, min_item_degree=5, device="cuda")
                r = torch.randint(1, 6, u.shape, device=u.device, generator=torch.Generator(device=u.device).manual_seed(1234))
                df = pd.DataFrame({"user_id": u.cpu().numpy(), "business_id": i.cpu().numpy(), "rating": r.cpu().numpy()})
            else:
                df = make_frame(mean_items=47.0, min_item_degree=5)
        else:
            nu, ni, mean = (float(x) for x in str(synthetic).split("x"))
            df = make_frame(int(nu), int(ni), mean)
        pipe._load_df = lambda: df
    args.data_pipeline = pipe
    if cfg.model_name == 'CDAE' and cfg.get("fast_loader"):
        # sparse device-side store instead of the dense pivot + four dense masks per user
        import torch
        from .data.cdae_batches import CDAEInteractions
        raw = pipe._load_df()
        nu, ni = int(raw.user_id.max()) + 1, int(raw.business_id.max()) + 1
        args.cdae_data = CDAEInteractions.from_interactions(
            torch.from_numpy(raw.user_id.values.astype('int64')), torch.from_numpy(raw.business_id.values.astype('int64')),
            nu, ni, seed=cfg.seed, device=cfg.device)
        args.model_info = {'num_items': ni, 'num_users': nu}
        return args
    df = pipe.preprocess()
    if cfg.model_name == 'CDAE':
        from .data.datasets.cdae_dataset import CDAEDataset
        train_data, valid_data, test_data = pipe.split(df)
        args.train_dataset = CDAEDataset(train_data, 'train', neg_times=cfg.neg_times)
        args.valid_dataset = CDAEDataset(valid_data, 'valid', neg_times=cfg.neg_times)
        args.test_dataset = CDAEDataset(test_data, 'test')
        args.model_info = {'num_items': len(df.columns) - 1, 'num_users': len(train_data)}
    else:
        train_data, valid_data, valid_eval_data, test_eval_data = pipe.split(df)
        args.train_dataset = MFDataset(train_data, num_items=pipe.num_items)
        args.valid_dataset = MFDataset(valid_data, num_items=pipe.num_items)
        args.valid_eval_data, args.test_eval_data = valid_eval_data, test_eval_data
        args.model_info = {'num_items': pipe.num_items, 'num_users': pipe.num_users}
    return args


def train(cfg, args):
    """reference train.py:74-115."""
    from .trainers.mf_trainer import MFTrainer
    from .trainers.ngcf_trainer import NGCFTrainer
    if cfg.get("fast_loader") and cfg.model_name in ('MF', 'NGCF'):
        # device-side epoch sampler instead of DataLoader(MFDataset): same distribution, own RNG
        from .data.triplets import EpochLoader
        import torch
        dev = torch.device(cfg.device)
        nu = args.model_info['num_users']
        train_dataloader = EpochLoader(args.train_dataset.to_sampler(dev, nu, seed=cfg.seed), cfg.batch_size, cfg.shuffle)
        valid_dataloader = EpochLoader(args.valid_dataset.to_sampler(dev, nu, seed=cfg.seed + 1), cfg.batch_size, cfg.shuffle)
    elif not (cfg.get("fast_loader") and cfg.model_name == 'CDAE'):
        train_dataloader = DataLoader(args.train_dataset, batch_size=cfg.batch_size, shuffle=cfg.shuffle)
        valid_dataloader = DataLoader(args.valid_dataset, batch_size=cfg.batch_size, shuffle=cfg.shuffle)
    if cfg.model_name == 'CDAE':
        from .trainers.cdae_trainer import CDAETrainer
        if cfg.get("fast_loader"):
            from .data.cdae_batches import CDAEBatchLoader
            from .engine import fused_eval_supports as engine_supports_topn
            # batches as lists straight from the per-user CSR (no dense rows or masks) when the fused step and the
            # fused evaluation can take them: NS-BCE, Adam / AdamW, a hidden size the kernels are built for
            as_lists = bool(cfg.get("list_batches", True)) and cfg.negative_sampling and cfg.get("fused_step", True) \
                and cfg.optimizer.lower() in ("adam", "adamw") and cfg.hidden_size in (16, 32, 64, 128) \
                and engine_supports_topn(cfg.top_n, cfg.hidden_size) and args.cdae_data.num_items <= 163840
            # the test metrics over list batches are computed for all users at once (CDAETrainer._scored_by_lists), so
            # the rows per evaluation batch do not enter the result: fewer, larger batches (7.1 -> 5.4 ms at Yelp2018 size)
            rows = lambda mode: max(cfg.batch_size, int(cfg.get("eval_batch_size", 4096))) if as_lists and mode == 'test' \
                else cfg.batch_size
            mk = lambda mode, seed: CDAEBatchLoader(args.cdae_data, mode, rows(mode), cfg.neg_times,
                                                    shuffle=cfg.shuffle and mode != 'test', seed=seed, lists=as_lists,
                                                    dropout=cfg.corruption_level)
            train_dataloader, valid_dataloader, test_dataloader = mk('train', cfg.seed), mk('valid', cfg.seed + 1), mk('test', 0)
        else:
            test_dataloader = DataLoader(args.test_dataset, batch_size=cfg.batch_size)
        trainer = CDAETrainer(cfg, args.model_info['num_items'], args.model_info['num_users'])
        trainer.run(train_dataloader, valid_dataloader)
        trainer.load_best_model()
        return trainer, trainer.evaluate(test_dataloader)
    if cfg.model_name == 'MF':
        trainer = MFTrainer(cfg, args.model_info['num_items'], args.model_info['num_users'])
    else:
        trainer = NGCFTrainer(cfg, args.model_info['num_items'], args.model_info['num_users'],
                              args.data_pipeline.laplacian_matrix)
    trainer.run(train_dataloader, valid_dataloader, args.valid_eval_data)
    trainer.load_best_model()
    return trainer, trainer.evaluate(args.test_eval_data, 'test')


def run(cfg, args):
    """reference train.py:56-60 (wandb init/finish omitted)."""
    set_seed(cfg.seed)
    return train(cfg, args)


def _init_distributed():
    """One process per GPU under ``python -m torch.distributed.run`` (MF only: the interaction
    matrix is then sharded by user, see trainers/mf_trainer.py); a plain launch stays single-GPU."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return False
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)              # RCCL over xGMI
    return True


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    distributed = _init_distributed()
    over = _parse_overrides(sys.argv[1:] if argv is None else argv)
    path = over.pop("config", None)
    model_name = over.pop("model_name", "MF")
    cfg = load_config(path, model_name=model_name, **over) if path else make_config(model_name, **over)
    args = build(cfg)
    trainer, metrics = run(cfg, args)
    logger.info(f"test precision/recall/map/ndcg @{cfg.top_n}: {metrics}")
    if distributed:
        import torch.distributed as dist
        dist.destroy_process_group()
    return metrics


if __name__ == '__main__':
    main()
