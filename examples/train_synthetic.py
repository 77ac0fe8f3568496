#!/usr/bin/env python3
"""The reference's training driver (train.py:52-100) on synthetic EBNeRD-shaped data, through this package's drop-in
Modules: model = UserModel(max_user_id) re-dimensioned as BASELINE's configs do, FlatAdam (= Adam(lr, wd=1e-5) as one
launch), host float64 batches staged one ahead, per-impression AUC on the device, a checkpoint without `delta` per epoch.

    python examples/train_synthetic.py --workload C1-demo --batches 20 --epochs 2
"""
import argparse
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from news_recommendation_model_amd import data_io, evaluation, synth, trainer  # noqa: E402
from news_recommendation_model_amd.config import Dims, WORKLOADS               # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="C1-demo", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--batches", type=int, default=20, help="batches per epoch (distinct seeds, re-used every epoch)")
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--ckpt-dir", default=None)
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise SystemExit("needs an MI355X: the Modules have no CPU path")
    wl = WORKLOADS[args.workload]
    B = args.batch or wl["B"]
    dims = Dims.for_emb(wl["emb"])
    user_num = 10 * B
    torch.manual_seed(args.seed)                                           # train.py:42-43
    model = trainer.build_model(dims, user_num, synth.make_state_dict(dims, seed=args.seed + 1, user_num=user_num))
    opt = trainer.FlatAdam(model, lr=args.lr)                              # train.py:48
    ckpt_dir = args.ckpt_dir or tempfile.mkdtemp(prefix="nrm_ckpt_")
    # the data takes the reference's route: records in zstd+pickle subvolumes behind a head file (process_data.py:252-291),
    # read back with load_processed_dataset (:92-145) and batched as DataLoader(shuffle=True) does (train.py:40)
    records = []
    for i in range(args.batches):
        records += data_io.records_from_batch(synth.make_batch(dims, B, wl["H"], wl["T"], seed=1000 + i, user_num=user_num))
    head = data_io.write_processed_dataset(records, os.path.join(ckpt_dir, "synthetic_train_processed"), subvolume_item_num=4 * B)
    records, max_user_id = data_io.load_processed_dataset(head)
    epoch_no = [0]

    def loader():
        epoch_no[0] += 1
        return (b for b in data_io.iter_batches(records, B, shuffle=True, seed=args.seed + epoch_no[0]) if len(b["user_id"]) == B)
    hosts = list(data_io.iter_batches(records, B, shuffle=False))
    hist = trainer.train_epochs(model, opt, loader, args.epochs,
                                ckpt_path=os.path.join(ckpt_dir, "ckpt_synthetic_epoch_{epoch}.pth"))
    for rec in hist:
        print("[epoch]:{epoch} [lr]:{lr:.3e} loss_avg={loss_avg:.4f} auc_avg={auc_avg:.4f} impressions={impressions}".format(**rec))
    # the checkpoint round-trips into a fresh model (test.py:159-160) and validates (verify.py:19-43)
    fresh = trainer.build_model(dims, user_num)
    evaluation.load_checkpoint(fresh, os.path.join(ckpt_dir, f"ckpt_synthetic_epoch_{args.epochs - 1}.pth"))
    auc, hit = evaluation.validate([fresh], (trainer.batch_to_device(h, "cuda") for h in hosts[:4]))
    # eval mode = BatchNorm running statistics (momentum 0.1): on this synthetic data (N(0,1) text/image vectors, un-normalised
    # history pooling) activations grow quickly while the model memorises, so after long runs the running statistics lag
    # the batch statistics and the eval-mode AUC can fall back to chance although the train-mode AUC keeps rising -- the
    # reference model behaves the same way (oracle/user_model_oracle.py reproduces it on the CPU)
    print(f"validation (eval mode) on 4 training batches: auc={auc:.4f} top1={hit:.4f}  checkpoints in {ckpt_dir}")


if __name__ == "__main__":
    main()
