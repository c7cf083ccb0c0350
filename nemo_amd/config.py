"""Recipe YAML -> plain dict, with the three OmegaConf behaviours the reference's ASR recipes rely on
(`examples/asr/conf/**/*.yaml`, loaded there by `@hydra_runner` -> `OmegaConf`):

  * `${a.b.c}` interpolations resolved against the root of the document (e.g. `feat_in: ${model.preprocessor.features}`,
    `d_model: ${model.encoder.d_model}`, `sample_rate: ${model.sample_rate}`);
  * YAML-1.2 numbers: `1e-3` (no dot) is a float, as in OmegaConf's loader -- PyYAML's YAML-1.1 resolver reads a string;
  * `???` marks a mandatory value: reading the config keeps it, `missing_keys()` lists what is still unset and
    `select(..., throw_on_missing=True)` raises like `omegaconf.errors.MissingMandatoryValue`.

Command-line style overrides (`model.tokenizer.dir=/x`, `+trainer.fast_dev_run=True`, `~model.spec_augment`) are applied
before interpolation, which is the order Hydra composes in.  No hydra / omegaconf import: both are absent on the build box
and optional for users."""
from __future__ import annotations

import copy
import re
from typing import Any, Dict, Iterable, List, Optional

import yaml

MISSING = "???"


class _Loader(yaml.SafeLoader):
    pass


# YAML 1.2 float (the resolver OmegaConf installs): digits with an exponent but no dot are numbers, not strings
_Loader.add_implicit_resolver(
    "tag:yaml.org,2002:float",
    re.compile(r"""^(?:[-+]?(?:[0-9][0-9_]*)\.[0-9_]*(?:[eE][-+]?[0-9]+)?
                   |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
                   |\.[0-9_]+(?:[eE][-+][0-9]+)?
                   |[-+]?\.(?:inf|Inf|INF)
                   |\.(?:nan|NaN|NAN))$""", re.X),
    list("-+0123456789."))

_INTERP = re.compile(r"\$\{([^${}]+)\}")


class MissingMandatoryValue(KeyError):
    pass


def _parse_scalar(text: str) -> Any:
    return yaml.load(text, Loader=_Loader) if text != "" else ""


def select(cfg: Dict[str, Any], dotted: str, throw_on_missing: bool = False) -> Any:
    node: Any = cfg
    for part in dotted.split("."):
        if isinstance(node, list):
            node = node[int(part)]
        else:
            node = node[part]
    if throw_on_missing and node == MISSING:
        raise MissingMandatoryValue(f"Missing mandatory value: {dotted}")
    return node


def _assign(cfg: Dict[str, Any], dotted: str, value: Any, create: bool) -> None:
    parts = dotted.split(".")
    node = cfg
    for part in parts[:-1]:
        if part not in node:
            if not create:
                raise KeyError(f"Could not override '{dotted}': no such key (use '+{dotted}=...' to add it)")
            node[part] = {}
        node = node[part]
    if parts[-1] not in node and not create:
        raise KeyError(f"Could not override '{dotted}': no such key (use '+{dotted}=...' to add it)")
    node[parts[-1]] = value


def apply_overrides(cfg: Dict[str, Any], overrides: Iterable[str]) -> Dict[str, Any]:
    for ov in overrides:
        if ov.startswith("~"):
            parts = ov[1:].split(".")
            node = cfg
            for part in parts[:-1]:
                node = node[part]
            node.pop(parts[-1])
            continue
        key, _, val = ov.partition("=")
        create = key.startswith("+")
        _assign(cfg, key.lstrip("+"), _parse_scalar(val), create)
    return cfg


def resolve(cfg: Dict[str, Any]) -> Dict[str, Any]:
    """replace every `${path}` by the (recursively resolved) value at `path`; a string that is exactly one interpolation takes
    the referenced value's type, otherwise the value is formatted into the string"""
    root = cfg

    def value_of(path: str, depth: int) -> Any:
        if depth > 32:
            raise RecursionError(f"interpolation cycle at ${{{path}}}")
        return res(select(root, path.strip()), depth + 1)

    def res(v: Any, depth: int = 0) -> Any:
        if isinstance(v, str):
            m = _INTERP.fullmatch(v)
            if m:
                return value_of(m.group(1), depth)
            return _INTERP.sub(lambda mm: str(value_of(mm.group(1), depth)), v)
        if isinstance(v, dict):
            return {k: res(x, depth) for k, x in v.items()}
        if isinstance(v, list):
            return [res(x, depth) for x in v]
        return v

    return res(copy.deepcopy(cfg))


def missing_keys(cfg: Any, prefix: str = "") -> List[str]:
    out: List[str] = []
    if isinstance(cfg, dict):
        for k, v in cfg.items():
            out += missing_keys(v, f"{prefix}{k}.")
    elif isinstance(cfg, list):
        for i, v in enumerate(cfg):
            out += missing_keys(v, f"{prefix}{i}.")
    elif cfg == MISSING:
        out.append(prefix[:-1])
    return out


def load_config(path: Optional[str] = None, text: Optional[str] = None, overrides: Iterable[str] = ()) -> Dict[str, Any]:
    if text is None:
        with open(path) as f:
            text = f.read()
    cfg = yaml.load(text, Loader=_Loader) or {}
    apply_overrides(cfg, overrides)
    return resolve(cfg)
