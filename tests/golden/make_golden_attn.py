#!/usr/bin/env python3
"""Golden vectors for the attention / decoder building blocks the ScoreMapModule is made of, produced by the REAL reference
classes `Attention`, `TransformerDecoderLayer(_scaled)`, `ContextDecoder(_Hierachical)` of /root/reference/models/_modified_BiomedCLIP.py
(:448-478, :520-590, :1194-1308).  Dev container only (needs /root/reference); writes data only: tests/golden/attn_golden.npz.

Import recipe.  The file cannot be imported plainly: `timm` is not installed and the module sits in the `models` package whose
__init__ pulls in the whole (partly missing) model tree.  It is therefore executed by path under a private package name, with
stand-in modules ONLY for imports the three classes never execute on their forward path:
  * timm.models.layers.drop_path                     -- used by DropPath (CLIP ViT blocks), never by the three classes
  * timm.models.layers.trunc_normal_                 -- called by ContextDecoder.__init__'s weight init only; mapped to
                                                        torch.nn.init.trunc_normal_ (same algorithm); every weight is then
                                                        OVERWRITTEN by the seeded state dict below, so init values never reach a
                                                        forward pass
  * timm.models.resnet.{ResNet,Bottleneck}, timm.models.vision_transformer.VisionTransformer  -- base classes / helpers of the CLIP
                                                        backbones, unused here
  * .BiomedCLIP.BiomedCLIP.hf_configs.arch_dict, .hf_model.{ClsPooler,_POOLERS}  -- HF text-encoder plumbing (:921-1003), unused here
`transformers` is the real installed package (imported before the stand-ins are installed).  Weights are not stored: generator and tests rebuild them from the same seeds (attn_fixture_util.seeded_state; torch's CPU
generator is platform-stable).  Dropout: the reference classes are
built with dropout=0 and put in eval() (the spec's sampling path).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_attn.py
"""
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from attn_fixture_util import ATTN_CASES, DEC_CASES, HIER_CASES, LAYER_SEED, SCALED_LAYER_SEED, seeded_state  # noqa: E402

REF = "/root/reference/models/_modified_BiomedCLIP.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "attn_golden.npz")
PKG = "_idiff_refmodels"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def load_reference():
    import transformers  # noqa: F401  the real package, before any stand-in exists
    import transformers.models.bert.modeling_bert  # noqa: F401

    def _unused(*a, **k):
        raise RuntimeError("stand-in reached: this import is not on the path of Attention/TransformerDecoderLayer/ContextDecoder")

    class _Base(torch.nn.Module):
        def __init__(self, *a, **k):
            _unused()

    if "timm" not in sys.modules:
        _mod("timm", __path__=[])
        _mod("timm.models", __path__=[])
        _mod("timm.models.layers", drop_path=_unused, trunc_normal_=torch.nn.init.trunc_normal_)
        _mod("timm.models.resnet", ResNet=_Base, Bottleneck=_Base)
        _mod("timm.models.vision_transformer", VisionTransformer=_Base)
    _mod(PKG, __path__=[])
    _mod(PKG + ".BiomedCLIP", __path__=[])
    _mod(PKG + ".BiomedCLIP.BiomedCLIP", __path__=[])
    _mod(PKG + ".BiomedCLIP.BiomedCLIP.hf_configs", arch_dict={})
    _mod(PKG + ".BiomedCLIP.BiomedCLIP.hf_model", ClsPooler=_Base, _POOLERS={})
    spec = importlib.util.spec_from_file_location(PKG + "._modified_BiomedCLIP", REF)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    out = {}
    g = torch.Generator().manual_seed(20260)

    with torch.no_grad():
        # ---- Attention (:448-478): self (N == M) and cross (few queries, many keys), 4 and 8 heads -------------------------
        for tag, (dim, heads, N, M, seed) in ATTN_CASES.items():
            m = ref.Attention(dim, heads, proj_drop=0.0).eval()
            m.load_state_dict(seeded_state(m, seed))
            q = torch.randn(2, N, dim, generator=g)
            kv = q if tag == "attn_self" else torch.randn(2, M, dim, generator=g)
            out[f"{tag}/q"], out[f"{tag}/kv"] = q.numpy(), kv.numpy()
            out[f"{tag}/out"] = m(q, kv, kv).numpy()
        # ---- TransformerDecoderLayer (:520-549) --------------------------------------------------------------------------
        m = ref.TransformerDecoderLayer(256, 4, dropout=0.0).eval()
        m.load_state_dict(seeded_state(m, LAYER_SEED))
        x = torch.randn(2, 5, 256, generator=g)
        mem = torch.randn(2, 144, 256, generator=g)
        out["layer/x"], out["layer/mem"], out["layer/out"] = x.numpy(), mem.numpy(), m(x, mem).numpy()
        # ---- ContextDecoder (:1194-1244): the spec's 3 layers at two visual widths, and the reference default of 6 ----------
        for tag, (layers, vdim, hw, seed) in DEC_CASES.items():
            m = ref.ContextDecoder(transformer_width=256, transformer_heads=4, transformer_layers=layers, visual_dim=vdim, text_dim=512,
                                   dropout=0.0).eval()
            m.load_state_dict(seeded_state(m, seed))
            text = torch.randn(2, 5, 512, generator=g)
            visual = torch.randn(2, hw, vdim, generator=g)
            out[f"{tag}/text"], out[f"{tag}/visual"] = text.numpy(), visual.numpy()
            out[f"{tag}/out"] = m(text, visual).numpy()
        # ---- appended after the cases above (their draws from `g`, hence their arrays, are unchanged) ---------------------------------
        # TransformerDecoderLayer_scaled (:552-590) and ContextDecoder_Hierachical (:1247-1308), if_flash=False (flash_attn is not
        # installed; Attention_flash has the same parameters) and if_scale=True (the only form the constructor accepts)
        m = ref.TransformerDecoderLayer_scaled(256, 4, dropout=0.0, if_flash=False).eval()
        m.load_state_dict(seeded_state(m, SCALED_LAYER_SEED))
        x = torch.randn(2, 5, 256, generator=g)
        mem = torch.randn(2, 144, 256, generator=g)
        out["slayer/x"], out["slayer/mem"], out["slayer/out"] = x.numpy(), mem.numpy(), m(x, mem).numpy()
        for tag, (layers, vdim, hw, outdim, seed) in HIER_CASES.items():
            m = ref.ContextDecoder_Hierachical(transformer_width=256, transformer_heads=4, transformer_layers=layers, visual_dim=vdim, text_dim=512,
                                               dropout=0.0, outdim=outdim, if_scale=True, if_flash=False).eval()
            m.load_state_dict(seeded_state(m, seed))
            text = torch.randn(2, 5, 512, generator=g)
            visual = torch.randn(2, hw, vdim, generator=g)
            out[f"{tag}/text"], out[f"{tag}/visual"] = text.numpy(), visual.numpy()
            out[f"{tag}/out"] = m(text, visual).numpy()
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, len(out), "arrays", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
