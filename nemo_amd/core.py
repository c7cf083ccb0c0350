"""Host-side mirror of the reference's NeuralModule contract for the hot path.

`NeuralModule = nn.Module + Typing + Serialization + FileIO` (nemo/core/classes/module.py:26).  When NeMo is installed
the drop-in classes can simply be named in a YAML `_target_`; this file keeps the same *behavioural* contract without
importing NeMo (hydra / omegaconf / lightning / wrapt are soft dependencies that are absent on the build box):

  * `input_types` / `output_types` ordered dicts of `NeuralType(axes, element_type)`       (core/neural_types/)
  * `@typecheck()`: kwargs-only calls, argument-name membership, ndim == len(axes), outputs tagged with `.neural_type`
    (core/classes/common.py:1011-1147: TypeError on violation); can be disabled globally like the reference
  * `from_config_dict` / `to_config_dict` with a `_target_` key and the reference's module-path aliases
    (core/classes/common.py:528-591)
  * `.nemo` save / restore: uncompressed tar of model_config.yaml + model_weights.ckpt (torch.save(state_dict))
    (core/connectors/save_restore_connector.py:49-91, 231-283).
"""
from __future__ import annotations

import functools
import importlib
import io
import os
import tarfile
import tempfile
from collections import OrderedDict
from typing import Any, Dict, Optional, Tuple

import torch
from torch import nn


class ElementType:
    def __repr__(self):
        return type(self).__name__


class AudioSignal(ElementType):
    def __init__(self, freq: Optional[int] = None):
        self.freq = freq


class LengthsType(ElementType): pass
class SpectrogramType(ElementType): pass
class MelSpectrogramType(SpectrogramType): pass
class AcousticEncodedRepresentation(ElementType): pass
class LogprobsType(ElementType): pass
class LabelsType(ElementType): pass
class LossType(ElementType): pass
class EmbeddedTextType(ElementType): pass
class ChannelType(ElementType): pass
class BoolType(ElementType): pass


class NeuralType:
    def __init__(self, axes: Optional[Tuple[str, ...]] = None, elements_type: Optional[ElementType] = None, optional=False):
        self.axes = tuple(axes) if axes is not None else None
        self.elements_type = elements_type
        self.optional = optional

    def __repr__(self):
        return f"NeuralType(axes={self.axes}, elements_type={self.elements_type})"


class typecheck:
    """Decorator with the reference's observable behaviour (core/classes/common.py:1011-1147)."""
    _enabled = True

    def __init__(self, input_types=None, output_types=None):
        self._in, self._out = input_types, output_types

    @classmethod
    def set_typecheck_enabled(cls, enabled: bool = True):
        cls._enabled = enabled

    def __call__(self, fn):
        deco = self

        @functools.wraps(fn)
        def wrapper(module, *args, **kwargs):
            if not typecheck._enabled:
                return fn(module, *args, **kwargs)
            in_types = deco._in if deco._in is not None else getattr(module, "input_types", None)
            out_types = deco._out if deco._out is not None else getattr(module, "output_types", None)
            if in_types is None:
                return fn(module, *args, **kwargs)
            if len(args) > 0:
                raise TypeError("All arguments must be passed by kwargs only for typed methods")  # common.py:1134-1135
            for k, v in kwargs.items():
                if k not in in_types:
                    raise TypeError(f"Input argument {k} has no corresponding input_type match. "
                                    f"Existing input_types = {list(in_types.keys())}")
                nt = in_types[k]
                if isinstance(v, torch.Tensor) and nt.axes is not None and v.dim() != len(nt.axes):
                    raise TypeError(f"Input shape mismatch occured for {k} in module {type(module).__name__} : "
                                    f"Input shape expected = {nt.axes} | Input shape found : {tuple(v.shape)}")
            mandatory = [k for k, t in in_types.items() if not t.optional]
            missing = [k for k in mandatory if k not in kwargs]
            if missing:
                raise TypeError(f"Number of input arguments provided ({len(kwargs)}) is < the number of mandatory "
                                f"arguments; missing {missing}")
            out = fn(module, **kwargs)
            if out_types:
                outs = out if isinstance(out, (tuple, list)) else (out,)
                for o, (name, nt) in zip(outs, out_types.items()):
                    if isinstance(o, torch.Tensor):
                        if nt.axes is not None and o.dim() != len(nt.axes):
                            raise TypeError(f"Output shape mismatch occured for {name} in module {type(module).__name__}")
                        try:
                            o.neural_type = nt
                        except Exception:
                            pass
            return out

        return wrapper


# ------------------------------------------------------------------------------------------------ real NeMo core, when installed
def _nemo_core():
    """(neural_types module, NeuralModule, typecheck) of an installed NeMo, else None.  With NeMo importable the drop-in
    modules ARE `nemo.core.classes.NeuralModule`s carrying the reference's own NeuralTypes, checked by the reference's own
    `typecheck` (core/classes/common.py:1011-1147) -- so they can be mixed freely with stock NeMo modules; the mirror above is
    the default, and the fallback for boxes without NeMo's dependencies (this build box).  OPT-IN with NEMO_AMD_NEMO_CORE=1: the
    real NeuralModule brings its own Serialization / Typing / FileIO mixins, which then precede this package's `Serialization` in
    the MRO (from_config_dict / to_config_dict resolve to the reference's), and that combination has only been exercised against
    a stand-in class here, never against an installed NeMo -- an integrator switches it on deliberately.
    A stub package (e.g. the import shim the oracle tooling uses) is recognised by the missing `_TYPECHECK_ENABLED` and ignored."""
    if os.environ.get("NEMO_AMD_NEMO_CORE", "0") != "1":
        return None
    try:
        common = importlib.import_module("nemo.core.classes.common")
        if not hasattr(common, "_TYPECHECK_ENABLED"):
            return None
        nt = importlib.import_module("nemo.core.neural_types")
        classes = importlib.import_module("nemo.core.classes")
        return nt, classes.NeuralModule, classes.typecheck
    except Exception:
        return None


_REAL = _nemo_core()
HAVE_NEMO_CORE = _REAL is not None
if HAVE_NEMO_CORE:
    _nt, _RefNeuralModule, typecheck = _REAL  # noqa: F811  (the reference's decorator replaces the mirror)
    NeuralType = _nt.NeuralType  # noqa: F811
    ElementType = _nt.ElementType  # noqa: F811
    for _name in ("AudioSignal", "LengthsType", "SpectrogramType", "MelSpectrogramType", "AcousticEncodedRepresentation",
                  "LogprobsType", "LabelsType", "LossType", "EmbeddedTextType", "ChannelType", "BoolType"):
        if hasattr(_nt, _name):
            globals()[_name] = getattr(_nt, _name)
    if not (isinstance(BoolType, type) and issubclass(BoolType, ElementType)):  # (a NeMo without BoolType: a stand-in of its base)
        BoolType = type("BoolType", (ElementType,), {})
    _ModuleBase = _RefNeuralModule
else:
    _ModuleBase = nn.Module


# module-path aliases so the reference's own YAML `_target_` strings resolve to the drop-in classes
TARGET_ALIASES = {
    "nemo.collections.asr.modules.AudioToMelSpectrogramPreprocessor": "nemo_amd.modules.AudioToMelSpectrogramPreprocessor",
    "nemo.collections.asr.modules.SpectrogramAugmentation": "nemo_amd.modules.SpectrogramAugmentation",
    "nemo.collections.asr.modules.ConformerEncoder": "nemo_amd.modules.ConformerEncoder",
    "nemo.collections.asr.modules.SqueezeformerEncoder": "nemo_amd.modules.SqueezeformerEncoder",
    "nemo.collections.asr.modules.ConvASRDecoder": "nemo_amd.modules.ConvASRDecoder",
    "nemo.collections.asr.losses.ctc.CTCLoss": "nemo_amd.modules.CTCLoss",
    "nemo.collections.asr.losses.CTCLoss": "nemo_amd.modules.CTCLoss",
    "nemo.collections.asr.models.EncDecCTCModel": "nemo_amd.models.EncDecCTCModel",
    "nemo.collections.asr.models.EncDecCTCModelBPE": "nemo_amd.models.EncDecCTCModelBPE",
    "nemo.collections.asr.models.ctc_bpe_models.EncDecCTCModelBPE": "nemo_amd.models.EncDecCTCModelBPE",
    "nemo.collections.asr.losses.rnnt.RNNTLoss": "nemo_amd.modules.RNNTLoss",
    "nemo.collections.asr.modules.RNNTDecoder": "nemo_amd.modules.RNNTDecoder",
    "nemo.collections.asr.modules.RNNTJoint": "nemo_amd.modules.RNNTJoint",
    "nemo.collections.asr.models.EncDecRNNTModel": "nemo_amd.models.EncDecRNNTModel",
    "nemo.collections.asr.models.EncDecRNNTBPEModel": "nemo_amd.models.EncDecRNNTModel",
}


def resolve_target(path: str):
    path = TARGET_ALIASES.get(path, path)
    if not path.startswith(("nemo_amd.", "torch.nn.")):
        raise ValueError(f"_target_ '{path}' is outside the allow-list (core/classes/common.py:61-99)")
    mod, _, name = path.rpartition(".")
    return getattr(importlib.import_module(mod), name)


class Serialization:
    @classmethod
    def from_config_dict(cls, config: Dict[str, Any], trainer=None):
        config = dict(config)
        target = config.pop("_target_", None)
        klass = resolve_target(target) if target else cls
        kwargs = {k: v for k, v in config.items() if not k.startswith("_")}
        inst = klass(**kwargs)
        inst._cfg = dict(config, _target_=target or f"{klass.__module__}.{klass.__name__}")
        return inst

    def to_config_dict(self) -> Dict[str, Any]:
        if getattr(self, "_cfg", None) is not None:
            return dict(self._cfg)
        raise NotImplementedError("to_config_dict() requires the module to be built with from_config_dict()")


class NeuralModule(_ModuleBase, Serialization):
    @property
    def input_types(self) -> Optional[Dict[str, NeuralType]]:
        return None

    @property
    def output_types(self) -> Optional[Dict[str, NeuralType]]:
        return None

    @property
    def num_weights(self):
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def unfreeze(self):
        for p in self.parameters():
            p.requires_grad = True
        self.train()


# ------------------------------------------------------------------------------------------------ .nemo files
MODEL_CONFIG_YAML = "model_config.yaml"
MODEL_WEIGHTS = "model_weights.ckpt"


def save_nemo(path: str, config: Dict[str, Any], state_dict: Dict[str, torch.Tensor],
              artifacts: Optional[Dict[str, str]] = None) -> None:
    """save_restore_connector.py:49-91: tar (uncompressed) of model_config.yaml + model_weights.ckpt.  `artifacts` maps a
    dotted config key (e.g. 'tokenizer.model_path') to a file: the file is packed as '<md5>_<basename>' and the config
    entry becomes 'nemo:<that name>' -- `ModelPT.register_artifact` (modelPT.py:218-270, save_restore_connector.py:400-520)"""
    import copy
    import hashlib

    import yaml

    config = copy.deepcopy(config)
    packed = []
    for key, src in (artifacts or {}).items():
        with open(src, "rb") as f:
            digest = hashlib.md5(f.read()).hexdigest()
        arc = f"{digest}_{os.path.basename(src)}"
        node = config
        parts = key.split(".")
        for k in parts[:-1]:
            node = node[k]
        node[parts[-1]] = "nemo:" + arc
        packed.append((src, arc))
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, MODEL_CONFIG_YAML), "w") as f:
            yaml.safe_dump(config, f)
        torch.save({k: v.detach().cpu() for k, v in state_dict.items()}, os.path.join(tmp, MODEL_WEIGHTS))
        with tarfile.open(path, "w:") as tar:
            tar.add(os.path.join(tmp, MODEL_CONFIG_YAML), arcname=MODEL_CONFIG_YAML)
            tar.add(os.path.join(tmp, MODEL_WEIGHTS), arcname=MODEL_WEIGHTS)
            for src, arc in packed:
                tar.add(src, arcname=arc)


_ARTIFACT_DIRS = []  # extracted artifacts live as long as the process (the tokenizer keeps its model file open lazily)


def load_nemo(path: str):
    """-> (config dict, state_dict); members are read by name (no path traversal: save_restore_connector.py:640).  Config
    values of the form 'nemo:<member>' are extracted to a private directory and replaced by the extracted path."""
    import yaml

    try:  # plain tar first; older .nemo checkpoints are gzip-compressed (save_restore_connector.py:684-694 does the same)
        tar = tarfile.open(path, "r:")
    except tarfile.ReadError:
        tar = tarfile.open(path, "r:gz")
    with tar:
        names = {os.path.basename(m.name): m for m in tar.getmembers() if m.isfile()}
        cfg = yaml.safe_load(tar.extractfile(names[MODEL_CONFIG_YAML]).read())
        sd = torch.load(io.BytesIO(tar.extractfile(names[MODEL_WEIGHTS]).read()), map_location="cpu", weights_only=True)
        out_dir = None

        def resolve(node):
            nonlocal out_dir
            items = node.items() if isinstance(node, dict) else enumerate(node) if isinstance(node, list) else ()
            for k, v in list(items):
                if isinstance(v, (dict, list)):
                    resolve(v)
                elif isinstance(v, str) and v.startswith("nemo:"):
                    member = os.path.basename(v[5:])
                    if member not in names:
                        raise FileNotFoundError(f"{path}: artifact '{member}' named by the config is not in the archive")
                    if out_dir is None:
                        out_dir = tempfile.mkdtemp(prefix="nemo_amd_artifacts_")
                        _ARTIFACT_DIRS.append(out_dir)
                    dst = os.path.join(out_dir, member)
                    with open(dst, "wb") as f:
                        f.write(tar.extractfile(names[member]).read())
                    node[k] = dst
        resolve(cfg)
    return cfg, sd
