"""YAML option handling with the reference's surface (options.py:19-143): `parse(opt_path, is_train)`,
`dict_to_nonedict`, `NoneDict`, `dict2str`, `check_resume`.  Same keys in, same derived keys out
(`is_train`, per-dataset `phase/scale/data_type`, `path.root/experiments_root/models/training_state/log/
val_images` or `path.results_root/log`), debug overrides, and CUDA_VISIBLE_DEVICES from `gpu_ids`
(only when not already pinned by a launcher: under torchrun each rank owns one device)."""
import logging
import os
import os.path as osp
from collections import OrderedDict

import yaml


def _ordered_loader():
    class Loader(yaml.SafeLoader):
        pass

    def construct(loader, node):
        loader.flatten_mapping(node)
        return OrderedDict(loader.construct_pairs(node))

    Loader.add_constructor(yaml.resolver.BaseResolver.DEFAULT_MAPPING_TAG, construct)
    return Loader


def parse(opt_path, is_train=True):
    with open(opt_path, mode="r") as f:
        opt = yaml.load(f, Loader=_ordered_loader())
    gpu_list = ",".join(str(x) for x in opt.get("gpu_ids", []))
    if gpu_list and "LOCAL_RANK" not in os.environ and "CUDA_VISIBLE_DEVICES" not in os.environ:
        os.environ["CUDA_VISIBLE_DEVICES"] = gpu_list  # options.py:23-24
    opt["is_train"] = is_train
    scale = 1
    for phase, dataset in (opt.get("datasets") or {}).items():
        dataset["phase"] = phase.split("_")[0]
        dataset["scale"] = scale
        is_lmdb = False
        for key in ("dataroot_GT", "dataroot_LQ"):
            if dataset.get(key) is not None:
                dataset[key] = osp.expanduser(dataset[key])
                is_lmdb = is_lmdb or dataset[key].endswith("lmdb")
        dataset["data_type"] = "lmdb" if is_lmdb else "img"
        if str(dataset.get("mode", "")).endswith("mc"):
            dataset["data_type"] = "mc"
            dataset["mode"] = dataset["mode"].replace("_mc", "")
    opt.setdefault("path", OrderedDict())
    for key, path in list(opt["path"].items()):
        if path and key != "strict_load" and isinstance(path, str):
            opt["path"][key] = osp.expanduser(path)
    root = opt["path"].get("root") or osp.abspath(osp.join(osp.dirname(osp.abspath(__file__)), osp.pardir))
    opt["path"]["root"] = root
    if is_train:
        exp = osp.join(root, "experiments", opt["name"])
        opt["path"].update(experiments_root=exp, models=osp.join(exp, "models"), training_state=osp.join(exp, "training_state"),
                           log=exp, val_images=osp.join(exp, "val_images"))
        if "debug" in opt["name"]:  # options.py:80-83
            opt["train"]["val_freq"] = 8
            opt["logger"]["print_freq"] = 1
            opt["logger"]["save_checkpoint_freq"] = 8
    else:
        res = osp.join(root, "results")
        opt["path"]["results_root"] = osp.join(res, opt["name"])
        opt["path"]["log"] = osp.join(res, opt["name"])
    return opt


def dict2str(opt, indent_l=1):
    msg = ""
    for k, v in opt.items():
        if isinstance(v, dict):
            msg += " " * (indent_l * 2) + k + ":[\n" + dict2str(v, indent_l + 1) + " " * (indent_l * 2) + "]\n"
        else:
            msg += " " * (indent_l * 2) + k + ": " + str(v) + "\n"
    return msg


class NoneDict(dict):
    def __missing__(self, key):
        return None


def dict_to_nonedict(opt):
    if isinstance(opt, dict):
        return NoneDict(**{k: dict_to_nonedict(v) for k, v in opt.items()})
    if isinstance(opt, list):
        return [dict_to_nonedict(v) for v in opt]
    return opt


def check_resume(opt, resume_iter):
    logger = logging.getLogger("base")
    if opt["path"]["resume_state"]:
        if opt["path"].get("pretrain_model_G") is not None or opt["path"].get("pretrain_model_D") is not None:
            logger.warning("pretrain_model path will be ignored when resuming training.")
        opt["path"]["pretrain_model_G"] = osp.join(opt["path"]["models"], "{}_G.pth".format(resume_iter))
        if "gan" in str(opt.get("model", "")):
            opt["path"]["pretrain_model_D"] = osp.join(opt["path"]["models"], "{}_D.pth".format(resume_iter))
