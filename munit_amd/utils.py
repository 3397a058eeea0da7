"""Host-side helpers of the hot path: config, init, LR schedule (reference: scripts/utils.py)."""
import math
import os

import torch
import torch.nn.init as init
import yaml
from torch.optim import lr_scheduler

ADAPTATION_DEFAULTS = dict(full_adaptation=0, output_classifier_lambda=0, output_adv_lambda=0,
                           output_classif_freq=1, adv_lambda=0, dfeat_lambda=0, classif_frequency=15,
                           sem_seg_lambda=0)


def get_config(config):
    """utils.py:743-758: yaml.safe_load + default optimizer 'adam'.  Additionally fills the
    `adaptation:` block with zeros when a (stale) config such as config_HD.yaml lacks it
     -- the reference would raise KeyError there."""
    with open(config, "r") as stream:
        conf = yaml.safe_load(stream)
    if "optimizer" not in conf:
        conf["optimizer"] = "adam"
    return normalize_config(conf)


def normalize_config(conf):
    conf.setdefault("optimizer", "adam")
    ad = dict(ADAPTATION_DEFAULTS)
    ad.update(conf.get("adaptation") or {})
    conf["adaptation"] = ad
    for k in ("semantic_w", "domain_adv_w", "vgg_w", "recon_synth_w"):
        conf.setdefault(k, 0)
    conf.setdefault("recon_mask", 0)
    return conf


def get_scheduler(optimizer, hyperparameters, iterations=-1):
    """utils.py:1066-1090."""
    if "lr_policy" not in hyperparameters or hyperparameters["lr_policy"] == "constant":
        scheduler = None
    elif hyperparameters["lr_policy"] == "step":
        if iterations != -1:
            for g in optimizer.param_groups:
                g.setdefault("initial_lr", g["lr"])
        scheduler = lr_scheduler.StepLR(optimizer, step_size=hyperparameters["step_size"],
                                        gamma=hyperparameters["gamma"], last_epoch=iterations)
    else:
        # the reference RETURNS the exception object here instead of raising it (utils.py:1087-1089); kept, because a
        # caller written against the reference sees the same value (and the same failure at the first .step())
        return NotImplementedError("learning rate policy [%s] is not implemented", hyperparameters["lr_policy"])
    return scheduler


def _fill_like_reference(param, fn):
    """Draw into a standard-contiguous CPU tensor (the layout the reference's parameters have, so
    the RNG stream maps to the same logical elements) and copy into the parameter's storage."""
    tmp = torch.empty(tuple(param.shape), dtype=torch.float32)
    fn(tmp)
    with torch.no_grad():
        param.copy_(tmp.to(param.device))


def weights_init(init_type="gaussian"):
    """utils.py:1093-1115: applies to modules whose class name starts with Conv / Linear and
    that own a `weight` (so the *Block wrappers are skipped exactly as in the reference)."""

    def init_fun(m):
        classname = m.__class__.__name__
        if (classname.find("Conv") == 0 or classname.find("Linear") == 0) and hasattr(m, "weight"):
            if init_type == "gaussian":
                _fill_like_reference(m.weight, lambda t: init.normal_(t, 0.0, 0.02))
            elif init_type == "xavier":
                _fill_like_reference(m.weight, lambda t: init.xavier_normal_(t, gain=math.sqrt(2)))
            elif init_type == "kaiming":
                _fill_like_reference(m.weight, lambda t: init.kaiming_normal_(t, a=0, mode="fan_in"))
            elif init_type == "orthogonal":
                _fill_like_reference(m.weight, lambda t: init.orthogonal_(t, gain=math.sqrt(2)))
            elif init_type == "default":
                pass
            else:
                assert 0, "Unsupported initialization: {}".format(init_type)
            if hasattr(m, "bias") and m.bias is not None:
                with torch.no_grad():
                    m.bias.zero_()

    return init_fun


def get_model_list(dirname, key):
    """Checkpoint discovery for resume (utils.py:887-908): among the regular files of `dirname`
    whose name contains both `key` and ".pt", the one that sorts last; None if the directory
    is missing or holds no match."""
    if not os.path.isdir(dirname):
        return None
    names = sorted(n for n in os.listdir(dirname)
                   if key in n and ".pt" in n and os.path.isfile(os.path.join(dirname, n)))
    return os.path.join(dirname, names[-1]) if names else None
