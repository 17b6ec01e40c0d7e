"""`CLIPDriftModel` -- the two-network (drift + noise) model wrapper with the reference's method surface
(models/drift_noise_model.py:27-756) over the MI355X-native nets.

Kept from the reference: constructor/`create_CLIPDriftModel` option names (:758-810), feed_data (:182-195),
optimize_parameters -> optimize_parameters_inputRes (:231-232, 242-312), optimize_score_map (:234-240),
test (:648-652), get_visuals/get_nets, save/load file naming `{iter}_{DP,NP,DN,NN}.pth` +
`lastest_*_ema.pth` [sic] (:670-755), save_training_state/resume_training, set_eval/set_train/set_gpu,
loss-message helpers, learning-rate helpers.
Changed on purpose:
  * data parallelism = ONE flat gradient buffer all-reduced over RCCL per step (parallel.FlatGradAllReduce)
    instead of 10 DistributedDataParallel wrappers with find_unused_parameters (:115-146);
  * the 9 per-step `.item()` host syncs (:299-309) become one device->host copy of a 5-float loss record;
  * the 8 dead alternative train steps (:314-629) are not reproduced (out of scope);
  * `.cuda()` is not hard-coded: the device is an argument;
  * the 224 hard-coded in optimize_score_map (:234) is the input size.
"""
import os
import time
from collections import OrderedDict

import torch
import torch.nn as nn

from .modules import create_net
from .modules.MSM_degEmb_Unet import ScoreMapModule
from .text_encoder import build_text_encoder


class EMA(nn.Module):
    """Minimal stand-in for ema_pytorch.EMA with the same state-dict key prefixes (`online_model.`, `ema_model.`)
    so `lastest_*_ema.pth` files keep their layout (drift_noise_model.py:122,139,151-152,687-692)."""

    def __init__(self, model, beta=0.995, update_every=10):
        super().__init__()
        import copy
        self.online_model = model
        self.ema_model = copy.deepcopy(model)
        for p in self.ema_model.parameters():
            p.requires_grad_(False)
        self.beta, self.update_every = beta, update_every
        self.register_buffer("step", torch.zeros((), dtype=torch.long))
        self.register_buffer("initted", torch.zeros((), dtype=torch.bool))

    @torch.no_grad()
    def update(self):
        self.step += 1
        if int(self.step) % self.update_every != 0:
            return
        if not bool(self.initted):
            for pe, po in zip(self.ema_model.parameters(), self.online_model.parameters()):
                pe.copy_(po)
            self.initted.fill_(True)
            return
        from .. import ops
        for pe, po in zip(self.ema_model.parameters(), self.online_model.parameters()):
            if pe.is_cuda and pe.is_contiguous() and po.is_contiguous():
                ops.axpby(pe.data, po.data, self.beta, 1.0 - self.beta, out=pe.data)  # HIP kernel, in place
            else:
                pe.lerp_(po.to(pe.dtype), 1.0 - self.beta)

    def forward(self, *a, **k):
        return self.ema_model(*a, **k)


class CLIPDriftModel():
    def __init__(self, text_encoder_pretrain_path, drift_net_lr, noise_net_lr, weight_decay_drift, beta1, beta2, nepoch, eta_min,
                 dist=False, gpu=True, optimize_type='predict_noise', optimize_target='std', if_train=True, dnet_settings=None,
                 nnet_settings=None, drift_loss='l2', noise_loss='none', if_MultiScoreMap=False, score_map_ch_mult=[1, 1, 2, 4],
                 score_map_ngf=64, use_image_context=False, use_degra_context=False, CLIP_Type="CLIP", device=None, text_encoder=None,
                 class_tokens=None, score_map_dropout=0.1, score_map_decoder="ContextDecoder", score_map_if_flash=False):
        """score_map_dropout: dropout of the ScoreMapModules' decoder blocks in training mode -- the reference builds them with
        ContextDecoder's default 0.1 (models/_modified_BiomedCLIP.py:1194-1201; drift_noise_model.py:110-112 passes no value);
        model option `score_map_dropout` overrides (0 = the deterministic training function of rounds 1-2).
        score_map_decoder: "ContextDecoder" (the frozen spec) or "ContextDecoder_Hierachical" (TransformerDecoderLayer_scaled blocks,
        models/_modified_BiomedCLIP.py:552-590,1247-1308); model option of the same name.
        score_map_if_flash (with ContextDecoder_Hierachical): the decoder attentions in the reference's half-precision form
        (Attention_flash, :481-517; the reference class's default) -- a labelled reduced-precision variant, inference only; False =
        fp32, what if_flash=False computes."""
        dnet_settings = dict(dnet_settings)
        nnet_settings = dict(nnet_settings)
        for s in (dnet_settings, nnet_settings):  # :58-61
            s['use_image_context'] = use_image_context
            s['use_degra_context'] = use_degra_context
        self.dnet_settings, self.nnet_settings = dnet_settings, nnet_settings
        self.score_map_ch_mult = score_map_ch_mult
        self.use_image_context, self.use_degra_context = use_image_context, use_degra_context
        self.optimize_target, self.optimize_type = optimize_target, optimize_type
        self.drift_loss, self.noise_loss = drift_loss, noise_loss
        self.dist = dist
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if (gpu and torch.cuda.is_available()) else torch.device("cpu")
        self.device = torch.device(device)
        if text_encoder is None:
            text_encoder, token_embed_dim = build_text_encoder(text_encoder_pretrain_path, CLIP_Type)
        else:
            token_embed_dim = 768 if CLIP_Type == "BiomedCLIP" else 512  # :70-77
        for p in text_encoder.parameters():
            p.requires_grad_(False)
        self.text_encoder = text_encoder.to(self.device)
        self.token_embed_dim = token_embed_dim
        self.score_map_dropout = float(score_map_dropout)

        def prompts(settings):
            if settings.get('text_module') != 'scoremap':
                return None
            if settings.get('if_MultiScoreMap'):
                return nn.ModuleList([ScoreMapModule(visual_dim=score_map_ngf * score_map_ch_mult[i], CLIP_Type=CLIP_Type,
                                                     token_embed_dim=token_embed_dim, dropout=score_map_dropout, decoder_type=score_map_decoder, if_flash=score_map_if_flash)
                                      for i in range(len(score_map_ch_mult))])
            # reference models/drift_noise_model.py:113-114,130-131: ONE default ScoreMapModule() handed to create_net().  The UNet that
            # consumes it is not part of the reference snapshot; this build's frozen spec puts it on the
            # full-resolution level (visual_dim = nf, the module's default 64) with its score map embedded into `score_map_chan` channels
            # of that level's skip
            nf = int(settings.get('nf', 64) or 64)
            return ScoreMapModule(visual_dim=nf, CLIP_Type=CLIP_Type, token_embed_dim=token_embed_dim, dropout=score_map_dropout,
                                  decoder_type=score_map_decoder, if_flash=score_map_if_flash)

        self.drift_prompt = prompts(dnet_settings)
        self.noise_prompt = prompts(nnet_settings)
        if class_tokens is not None:  # real class-prompt token ids [K, N1] (the reference: clip.tokenize(prompts, context_length=N1))
            for smms in (self.drift_prompt, self.noise_prompt):
                for m in ([] if smms is None else (smms if isinstance(smms, nn.ModuleList) else [smms])):
                    m.set_class_tokens(class_tokens)
        self.drift_net = create_net(dnet_settings, CLIP_ScoreMapModule=self.drift_prompt).to(self.device)
        self.noise_net = create_net(nnet_settings, CLIP_ScoreMapModule=self.noise_prompt).to(self.device)
        if self.drift_prompt is not None:
            self.dp_ema = EMA(self.drift_prompt, beta=0.995, update_every=10)
            self.np_ema = EMA(self.noise_prompt, beta=0.995, update_every=10)
        self.dn_ema = self.nn_ema = None  # created lazily: a deep copy of both nets doubles weight memory (reference :151-152)
        self.grad_sync = None
        if dist and torch.distributed.is_available() and torch.distributed.is_initialized():
            from ..parallel import GradSync
            self.grad_sync = GradSync()
            self.grad_sync.broadcast_parameters(list(self.drift_net.parameters()) + list(self.noise_net.parameters()))
        if if_train:
            from ..train_ops import FusedAdam
            self.drift_optimizer = FusedAdam(self.drift_net.parameters(), lr=drift_net_lr, weight_decay=weight_decay_drift, betas=(beta1, beta2))
            self.noise_optimizer = FusedAdam(self.noise_net.parameters(), lr=noise_net_lr, weight_decay=weight_decay_drift, betas=(beta1, beta2))
            self.drift_lr_scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.drift_optimizer, T_max=nepoch, eta_min=eta_min)
            self.noise_lr_scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.noise_optimizer, T_max=nepoch, eta_min=eta_min)
        self.visuals = None
        self.reinit_loss_message()

    # ---- reference plumbing ------------------------------------------------------------------------
    def update_lr(self):
        self.drift_lr_scheduler.step()
        self.noise_lr_scheduler.step()

    def get_current_learning_rate(self):
        return self.noise_optimizer.param_groups[0]["lr"]

    def set_sde(self, sde):
        self.sde = sde

    def feed_data(self, data):  # :182-195
        self.input = data['input'].to(self.device, torch.float32).contiguous()
        self.target = data['target'].to(self.device, torch.float32).contiguous()
        self.names = list(data['names'])
        self.A_emb = data['A_emb'].to(self.device, torch.float32).contiguous() if self.use_image_context else None
        time_idx, drift_noised_x, drift, std_noise, noise = self.sde.forward_diffusion(self.target, self.input)
        self.t, self.drift_noised_x, self.drift, self.std_noise, self.noise = time_idx, drift_noised_x, drift, std_noise, noise

    def reinit_loss_message(self):  # :197-218
        self.loss_info = {'latest': {'l': 0, 'nsml': 0, 'dsml': 0, 'nl': 0, 'dl': 0},
                          'avg': {'l': 0, 'dl': 0, 'nl': 0, 'dsml': 0, 'nsml': 0}, 'num': 0}

    def get_loss_message(self):  # :220-229
        num = max(self.loss_info['num'], 1)
        message = ""
        for k in self.loss_info['latest'].keys():
            message += '({}={:4f}/{:4f})'.format(k, self.loss_info['latest'][k], self.loss_info['avg'][k] / num)
        return message

    def optimize_parameters(self):  # :231-232
        return self.optimize_parameters_inputRes()

    def optimize_score_map(self, score_maps, label, size=None, mult=[1, 2, 4, 8], want_grads=False):
        """sum_i MSE(score_maps[i], Resize((size[0]//mult[i], size[1]//mult[i]))(label)) / 2   (:234-240).
        `size` defaults to the label's own H, W (the reference hard-codes [224, 224], its dataset's size).  Resize is torchvision's
        bilinear tensor resize WITHOUT antialiasing: the reference pins PyTorch 1.13.1 (README.md:28), i.e. torchvision 0.14, where
        `Resize(size)` on a tensor means antialias=None = off (it became on-by-default for tensors only in 0.17).
        Returns the loss as a 0-dim device tensor (HIP resize + MSE kernels; no graph is recorded -- the training step feeds
        autograd the explicit per-map gradients, which `want_grads=True` also returns)."""
        from ..train_ops import score_map_losses
        label = label.contiguous()
        H, W = (int(size[0]), int(size[1])) if size is not None else tuple(label.shape[-2:])
        rec = torch.zeros(len(score_maps), device=label.device, dtype=torch.float32)
        grads = score_map_losses(score_maps, label, rec, 0, mult=mult, size=(H, W), want_grad=want_grads)
        loss = rec.sum() / 2.0
        return (loss, grads) if want_grads else loss

    def optimize_parameters_inputRes(self):
        """(x_t-cond, cond)->drift; (x_t-cond, x_t)->noise; L2 + score-map pyramid losses; Adam (:242-312)."""
        from ..train_ops import train_step_inputRes
        return train_step_inputRes(self)

    # ---- sampling ----------------------------------------------------------------------------------
    @torch.no_grad()
    def test(self, **kw):  # :648-652
        out = self.sde.reverse_ddpm(self.input, self.names, self.text_encoder, reverse_type=self.optimize_target,
                                    optimize_type=self.optimize_type, image_context=self.A_emb, **kw)
        self.output = out
        self.visuals = out.detach().cpu().numpy()

    def get_visuals(self):
        return self.visuals

    def get_nets(self, use_ema=False):  # :657-668
        if use_ema:
            self._ensure_net_ema()
            return {'noise_net': self.nn_ema, 'drift_net': self.dn_ema}
        return {'noise_net': self.noise_net, 'drift_net': self.drift_net}

    def _ensure_net_ema(self):
        if self.dn_ema is None:
            self.dn_ema = EMA(self.drift_net, beta=0.995, update_every=10)
            self.nn_ema = EMA(self.noise_net, beta=0.995, update_every=10)

    def set_eval(self):
        self.drift_net.eval()
        self.noise_net.eval()

    def set_train(self):
        self.drift_net.train()
        self.noise_net.train()

    def set_gpu(self, device):  # :635-642
        self.drift_net.to(device)
        self.noise_net.to(device)
        self.text_encoder = self.text_encoder.to(device)
        self.device = torch.device(device)

    # ---- checkpoint wire format (:670-755) -----------------------------------------------------------
    @staticmethod
    def save_network(network, network_label, iter_label, save_dir):
        state_dict = OrderedDict((k, v.detach().cpu()) for k, v in network.state_dict().items())
        torch.save(state_dict, os.path.join(save_dir, "{}_{}.pth".format(iter_label, network_label)))

    def save(self, iter_label, save_dir):
        os.makedirs(save_dir, exist_ok=True)
        self._ensure_net_ema()
        if self.dnet_settings['text_module'] == 'scoremap':
            self.save_network(self.drift_prompt, "DP", iter_label, save_dir)
            self.save_network(self.noise_prompt, "NP", iter_label, save_dir)
            self.save_network(self.dp_ema, "DP_ema", 'lastest', save_dir)
            self.save_network(self.np_ema, "NP_ema", 'lastest', save_dir)
        self.save_network(self.drift_net, "DN", iter_label, save_dir)
        self.save_network(self.noise_net, "NN", iter_label, save_dir)
        self.save_network(self.dn_ema, "DN_ema", 'lastest', save_dir)
        self.save_network(self.nn_ema, "NN_ema", 'lastest', save_dir)

    def save_training_state(self, epoch, iter_step, save_dir):
        os.makedirs(save_dir, exist_ok=True)
        state = {"epoch": epoch, "iter": iter_step,
                 "schedulers": [self.drift_lr_scheduler.state_dict(), self.noise_lr_scheduler.state_dict()],
                 "optimizers": [self.drift_optimizer.state_dict(), self.noise_optimizer.state_dict()]}
        torch.save(state, os.path.join(save_dir, "{}.state".format(iter_step)))

    @staticmethod
    def load_training_state(path, trusted=False):
        """Read a `{iter}.state` file (the call trainUM.py makes before resume_training).  This build's layout (state dicts) loads
        under torch's `weights_only=True`.  The reference pickles the optimizer and scheduler OBJECTS (:694-704), which that mode
        rejects (it cannot even rebuild Adam's defaultdict); they are read by an unpickler whose find_class admits exactly what such
        a file holds -- torch.optim.Adam, CosineAnnealingLR, Parameter / tensor / storage rebuilders, defaultdict / OrderedDict --
        and nothing that could run code.  A pickle holding anything else needs `trusted=True` (option path.resume_state_trusted),
        i.e. the unrestricted unpickler the reference's torch 1.13 `torch.load` was."""
        import pickle
        try:
            return torch.load(path, map_location="cpu", weights_only=True)
        except pickle.UnpicklingError:
            pass
        try:
            return torch.load(path, map_location="cpu", weights_only=False, pickle_module=_restricted_pickle())
        except pickle.UnpicklingError as e:
            if not trusted:
                raise RuntimeError(f"{path}: holds more than state dicts and pickled Adam / CosineAnnealingLR objects ({e}); set "
                                   f"path.resume_state_trusted: true to unpickle it without restrictions") from e
        return torch.load(path, map_location="cpu", weights_only=False)

    def resume_training(self, resume_state):
        """Both `.state` layouts resume: this build's (state dicts) and the reference's, which pickles the scheduler and optimizer
        OBJECTS (models/drift_noise_model.py:694-704; its own resume_training then swaps them in wholesale, :702-704 -- optimizers
        that update orphaned copies of the weights).  From reference objects the Adam moments, step counts, hyper-parameters and
        the schedulers' epochs are taken over into the live fused optimizers."""
        for sch, s in zip((self.drift_lr_scheduler, self.noise_lr_scheduler), resume_state['schedulers']):
            if hasattr(s, "state_dict"):  # a pickled scheduler object
                s = {k: v for k, v in s.state_dict().items() if k != "optimizer"}
            sch.load_state_dict(s)
        for opt, s in zip((self.drift_optimizer, self.noise_optimizer), resume_state['optimizers']):
            if hasattr(s, "state_dict") or "flat" not in s:  # a torch.optim.Adam object or its state dict
                opt.load_torch_adam(s)
            else:
                opt.load_state_dict(s)

    @staticmethod
    def load_network(load_path, network, strict=True):
        load_net = torch.load(load_path, map_location="cpu")
        clean = OrderedDict()
        for k, v in load_net.items():  # strip DDP prefixes of reference-era checkpoints (:712-730)
            clean[k.replace('module.', '')] = v
        network.load_state_dict(clean, strict=strict)

    def load(self, iter_label, save_dir):
        if self.dnet_settings['text_module'] == 'scoremap':
            self.load_network(os.path.join(save_dir, f"{iter_label}_DP.pth"), self.drift_prompt)
            self.load_network(os.path.join(save_dir, f"{iter_label}_NP.pth"), self.noise_prompt)
        self.load_network(os.path.join(save_dir, f"{iter_label}_DN.pth"), self.drift_net)
        self.load_network(os.path.join(save_dir, f"{iter_label}_NN.pth"), self.noise_net)
        self._ensure_net_ema()
        for label, net in (("DP", getattr(self, "dp_ema", None)), ("NP", getattr(self, "np_ema", None)), ("DN", self.dn_ema), ("NN", self.nn_ema)):
            path = os.path.join(save_dir, f"lastest_{label}_ema.pth")
            if net is not None and os.path.exists(path):
                self.load_network(path, net)


def _restricted_pickle():
    """a `pickle`-shaped module for torch.load(pickle_module=...) whose Unpickler resolves only the globals of a reference-era
    `.state` file (see CLIPDriftModel.load_training_state)"""
    import pickle
    import types
    allowed = {("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "dict"), ("builtins", "set"),
               ("builtins", "list"), ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"),
               ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_parameter"), ("torch._utils", "_rebuild_parameter_with_state"),
               ("torch._tensor", "_rebuild_from_type_v2"), ("torch.nn.parameter", "Parameter"), ("torch", "Size"), ("torch", "device"),
               ("torch.storage", "UntypedStorage"), ("torch.serialization", "_get_layout"),
               ("torch.optim.adam", "Adam"), ("torch.optim.lr_scheduler", "CosineAnnealingLR")}

    class Unpickler(pickle.Unpickler):
        def find_class(self, module, name):
            key = ("builtins" if module == "__builtin__" else module, name)  # protocol-2 pickles (torch.save's default) say __builtin__
            if key in allowed or (module == "torch" and name.endswith("Storage")):
                return super().find_class(module, name)
            raise pickle.UnpicklingError(f"global {module}.{name} is not on the .state allow-list")

    # load / loads go through the allow-listing class too: torch's legacy (non-zip) loader calls pickle_module.load(f) directly for
    # the magic number, protocol, sys_info and storage keys -- with the plain pickle.load a crafted single-pickle file ran its
    # __reduce__ callable before any find_class of ours was asked
    def load(f, **kw):
        return Unpickler(f, **kw).load()

    def loads(b, **kw):
        import io
        return Unpickler(io.BytesIO(b), **kw).load()

    mod = types.ModuleType("idiff_restricted_pickle")
    mod.__dict__.update({k: getattr(pickle, k) for k in ("dump", "dumps", "Pickler", "PickleError", "UnpicklingError",
                                                         "HIGHEST_PROTOCOL", "DEFAULT_PROTOCOL")})
    mod.load, mod.loads = load, loads
    mod.Unpickler = Unpickler
    return mod


def create_CLIPDriftModel(train_opt, model_opt, phase='train', **extra):  # :758-810
    kw = dict(drift_net_lr=model_opt['drift_net_lr'], noise_net_lr=model_opt['noise_net_lr'],
              weight_decay_drift=model_opt['weight_decay_drift'], beta1=model_opt['beta1'], beta2=model_opt['beta2'],
              nepoch=train_opt['nepoch'], eta_min=model_opt['eta_min'], optimize_target=model_opt['optimize_target'],
              optimize_type=model_opt['optimize_type'], if_train=(phase == 'train'), dnet_settings=model_opt['dnet_settings'],
              nnet_settings=model_opt['nnet_settings'], drift_loss=model_opt['drift_loss'], noise_loss=model_opt['noise_loss'],
              dist=train_opt['dist'], use_image_context=model_opt['use_image_context'], use_degra_context=model_opt['use_degra_context'],
              CLIP_Type=model_opt["CLIP_Type"] if "CLIP_Type" in model_opt and model_opt["CLIP_Type"] else "CLIP")
    if 'if_MultiScoreMap' in model_opt and model_opt['if_MultiScoreMap'] is not None:
        kw.update(if_MultiScoreMap=model_opt['if_MultiScoreMap'], score_map_ch_mult=model_opt['score_map_ch_mult'],
                  score_map_ngf=model_opt['score_map_ngf'])
    if model_opt.get('score_map_dropout') is not None:
        kw.update(score_map_dropout=float(model_opt['score_map_dropout']))
    if model_opt.get('score_map_decoder'):
        kw.update(score_map_decoder=str(model_opt['score_map_decoder']))
    if model_opt.get('score_map_if_flash') is not None:
        kw.update(score_map_if_flash=bool(model_opt['score_map_if_flash']))
    kw.update(extra)
    return CLIPDriftModel(model_opt['text_encoder_pretrain_path'], **kw)
