#!/usr/bin/env python3
"""Training driver with the reference's CLI and option surface (trainUM.py:81-359):

    python -m instancediff_amd.trainUM -opt <yaml> [--launcher none|pytorch] [--local_rank N]
    python -m torch.distributed.run --nproc-per-node N -m instancediff_amd.trainUM -opt <yaml> --launcher pytorch

Differences from the reference (all in SURVEY.md §2.1/§3.1): world size comes from the environment (the
reference hard-codes 2, :66); one flat RCCL gradient all-reduce per optimizer per step instead of 10 DDP wrappers;
validation metrics (RMSE/PSNR/SSIM, :314-329) are computed on the device by one kernel; `max_iters` (optional
key under `train:`) bounds a run for smoke tests.
"""
import argparse
import math
import os
import random
import sys

import numpy as np
import torch

from . import options as option
from . import ops, parallel
from .data import DistIterSampler, create_dataset, dump_raw, iterate_batches
from .models import create_model
from .models.SDEs import create_sde


def set_seed(seed=1):  # trainUM.py:73-78
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


def validate(model, val_set, out_dir, limit=10):
    """first `limit` validation images: sampling + on-device metrics (:287-348)"""
    model.set_eval()
    acc = np.zeros(3)
    n = 0
    with torch.no_grad():
        for jj, vd in enumerate(iterate_batches(val_set, 1)):
            data = {'input': vd["LQ"], 'target': vd["GT"], 'names': vd["name"], 'A_emb': vd["A_emb"]}
            model.feed_data(data)
            model.test()
            m = ops.image_metrics(model.output[:, 0], model.target[:, 0]).cpu().numpy()[0]
            acc += m
            n += 1
            dump_raw(os.path.join(out_dir, f"{jj}_.raw"), vd["LQ"].numpy(), model.get_visuals(), vd["GT"].numpy())
            if n >= limit:
                break
    model.set_train()
    return acc / max(n, 1)


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-opt", type=str, help="Path to option YAML file.")
    parser.add_argument("--launcher", choices=["none", "pytorch"], default="none", help="job launcher")
    parser.add_argument("--local_rank", type=int, default=0)
    args = parser.parse_args(argv)
    opt = option.dict_to_nonedict(option.parse(args.opt, is_train=True))
    set_seed(opt["train"]["manual_seed"])
    if args.launcher == "none":
        opt["dist"] = False
        rank, world = -1, 1
    else:
        opt["dist"] = True
        rank, world, local = parallel.init_distributed()
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))

    train_set = val_set = None
    for phase, dataset_opt in opt["datasets"].items():
        if phase == "train":
            train_set = create_dataset(dataset_opt)
            bs = dataset_opt["batch_size"]
            if opt["dist"]:
                assert bs % world == 0  # data/__init__.py:14
                bs //= world
            train_bs = bs
        elif phase == "val":
            val_set = create_dataset(dataset_opt)
    assert train_set is not None and val_set is not None
    train_size = int(math.ceil(len(train_set) / opt["datasets"]["train"]["batch_size"]))
    total_epochs = opt["train"]["nepoch"]
    sampler = DistIterSampler(train_set, world, max(rank, 0), 1) if opt["dist"] else None
    if rank <= 0:
        print(f"Number of train images: {len(train_set)}, iters: {train_size}; total epochs: {total_epochs}")
        for k in ("models", "training_state", "val_images"):
            os.makedirs(opt["path"][k], exist_ok=True)

    train_opt = dict(opt["train"])
    train_opt["dist"] = opt["dist"]
    model = create_model(train_opt, opt["models"][train_opt["which_model"]])
    current_step, start_epoch = 0, 0
    if opt["path"]["resume_state"]:
        # reference-era .state files pickle optimizer / scheduler objects (drift_noise_model.py:694-704): see load_training_state
        resume_state = model.load_training_state(opt["path"]["resume_state"], trusted=bool(opt["path"].get("resume_state_trusted")))
        option.check_resume(opt, resume_state["iter"])
        start_epoch, current_step = resume_state["epoch"] + 1, resume_state["iter"]
        model.resume_training(resume_state)
        model.load(current_step, opt["path"]["models"])
    sde = create_sde(model.get_nets(), opt["sdes"][train_opt["which_sde"]])
    sde.set_gpu(model.device)
    model.set_sde(sde)
    max_iters = opt["train"]["max_iters"] or 0

    print("Start training from epoch: {:d}, iter: {:d}".format(start_epoch, current_step))
    done = False
    for epoch in range(start_epoch, total_epochs + 1):
        if sampler is not None:
            sampler.set_epoch(epoch)
        model.reinit_loss_message()
        for ii, td in enumerate(iterate_batches(train_set, train_bs, sampler=sampler, shuffle=True, seed=epoch)):
            current_step += 1
            data = {'input': td["LQ"], 'target': td["GT"], 'names': td["name"], 'A_emb': td["A_emb"]}
            model.feed_data(data)
            loss, dur = model.optimize_parameters()
            message = "<epoch:{:3d}, iter:{:8,d}, lr:{:.3e}> (fwd time {:.4f}) ".format(epoch, current_step, model.get_current_learning_rate(), dur)
            message += model.get_loss_message()
            if current_step % opt["logger"]["print_freq"] == 0 and rank <= 0:
                print(message)
            if current_step % opt["logger"]["save_checkpoint_freq"] == 0 and rank <= 0:
                model.save(current_step, opt["path"]["models"])
                model.save_training_state(epoch, current_step, opt["path"]["training_state"])
            if current_step % opt["train"]["val_freq"] == 0 and rank <= 0:
                rmse, psnr, ssim = validate(model, val_set, opt["path"]["val_images"])
                print("<epoch:{:3d}, iter:{:8,d}> # Validation # PSNR: {:.6f} # SSIM: {:.6f} # RMSE: {:.6f}".format(epoch, current_step, psnr,
                                                                                                                   ssim, rmse))
            if max_iters and current_step >= max_iters:
                done = True
                break
        if done:
            break
        if epoch % 5 == 0 and rank <= 0:
            model.save(f"epoch_{epoch}", opt["path"]["models"])
            model.save_training_state(epoch, current_step, opt["path"]["training_state"])
    if rank <= 0:
        model.save("latest", opt["path"]["models"])
        print("End of training.")
    if opt["dist"] and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return current_step


if __name__ == "__main__":
    main()
