#!/usr/bin/env python3
"""Inference driver with the reference's option surface (testUM.py:43-186):

    python -m instancediff_amd.testUM -opt <yaml>

Reads `test: {iter, pth_dir, use_ema, which_model, which_sde, result_root}` (testUM.py:71-100), restores every
image of the test sets with `model.test()` (timed like :141-144), computes RMSE / PSNR / SSIM per image on the
device (:151-164) and writes the LQ|pred|GT `.raw` triptychs (:170-173).  With `--random-init` the checkpoint load
is skipped (synthetic smoke runs).  Sampling shards by image across ranks when launched with torchrun.
"""
import argparse
import os
import time
from collections import OrderedDict

import torch
import yaml

from . import ops, parallel
from .data import create_dataset, dump_raw, iterate_batches
from .models import create_model
from .models.SDEs import create_sde


def main(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-opt", type=str, required=True, help="Path to options YAML file.")
    parser.add_argument("--random-init", action="store_true", help="skip model.load (no checkpoint; synthetic smoke run)")
    parser.add_argument("--limit", type=int, default=0, help="stop after N images")
    args = parser.parse_args(argv)
    with open(args.opt, "r") as f:
        opt = yaml.load(f.read(), yaml.FullLoader)  # raw dict: missing keys raise, as in the reference (:50-54)
    rank, world, local = parallel.init_distributed()
    if torch.cuda.is_available():
        torch.cuda.set_device(local if world > 1 else opt['gpu_ids'][0])
    train_opt, test_opt = dict(opt['train']), opt['test']
    train_opt['dist'] = False
    model = create_model(train_opt, opt['models'][test_opt['which_model']], phase='test')
    if not args.random_init:
        model.load(test_opt['iter'], test_opt['pth_dir'])
    sde = create_sde(model.get_nets(use_ema=test_opt['use_ema']), opt['sdes'][test_opt['which_sde']])
    sde.set_gpu(model.device)
    model.set_sde(sde)
    model.set_eval()
    result_root = os.path.join(test_opt['result_root'], opt['name'])
    results = OrderedDict((a, {'num': 0, 'RMSE': [], 'SSIM': [], 'PSNR': []}) for a in opt['artifact_type'])
    times = []
    n_done = 0
    for phase, dataset_opt in sorted(opt["datasets"].items()):
        if args.limit and n_done >= args.limit:
            break
        dataset_opt = dict(dataset_opt)
        dataset_opt.setdefault("phase", phase.split("_")[0])
        test_set = create_dataset(dataset_opt)
        ids = parallel.shard_indices(len(test_set), rank, world)
        print("Testing [{:s}]: {:d} images ({:d} on this rank)".format(dataset_opt["name"], len(test_set), len(ids)))
        with torch.no_grad():
            for i in ids:
                it = test_set[i]
                if it["name"] not in opt['artifact_type']:
                    continue
                data = {'input': it["LQ"][None], 'target': it["GT"][None], 'names': [it["name"]], 'A_emb': it["A_emb"][None]}
                model.feed_data(data)
                torch.cuda.synchronize()
                tic = time.time()
                model.test()
                times.append(time.time() - tic)
                rmse, psnr, ssim = ops.image_metrics(model.output[:, 0], model.target[:, 0]).cpu().tolist()[0]
                r = results[it["name"]]
                r['RMSE'].append(rmse), r['SSIM'].append(ssim), r['PSNR'].append(psnr)
                r['num'] += 1
                shape = dump_raw(os.path.join(result_root, it["name"], f"{i}.raw"), it["LQ"].numpy(), model.get_visuals(), it["GT"].numpy())
                os.replace(os.path.join(result_root, it["name"], f"{i}.raw"),
                           os.path.join(result_root, it["name"], f"{i}_{shape[-1]}x{shape[-2]}x1.raw"))
                print(f' Testing {i}, {it["GT_path"]}: RMSE={rmse}, SSIM={ssim}, PSNR={psnr}')
                n_done += 1
                if args.limit and n_done >= args.limit:
                    break
    for k, v in results.items():
        if v['num']:
            print(k + "".join(f", AVG {m}: {sum(v[m]) / v['num']}" for m in ('RMSE', 'SSIM', 'PSNR')))
    if times:
        print(f"mean sampling time per image: {sum(times) / len(times):.3f} s ({sde.T} steps)")
    if world > 1 and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
    return results


if __name__ == "__main__":
    main()
