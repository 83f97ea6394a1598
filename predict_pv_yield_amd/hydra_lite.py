"""A small Hydra/OmegaConf-compatible composer (hydra-core / omegaconf are not installed in this image).

Covers exactly the features the reference's config tree uses (SURVEY.md §5 "Config / flags"):
  * defaults-list composition of config groups (configs/config.yaml:4-15), `null` entries, optional `.yaml`
  * `# @package _global_` experiment overlays with `override /group: file` (configs/experiment/conv3d.yaml:6-11)
  * command-line overrides: `group=file`, `a.b.c=value`, `+a.b=value`, `~a.b`
  * interpolation: `${a.b}`, `${hydra:runtime.cwd}`, `${oc.env:VAR[,default]}`, `${now:%Y-%m-%d}`
  * `instantiate(cfg, **kwargs)` of `_target_` dotted paths (`_convert_`, `_recursive_` accepted), with an
    alias table so the REFERENCE's own yaml files (`predict_pv_yield.models...`, `pytorch_lightning.Trainer`,
    `nowcasting_dataloader.datamodules.NetCDFDataModule`) resolve to this package when the originals are absent
  * `compose(config_dir, config_name, overrides)` + `initialize`-free usage from tests, and a `main` decorator
    that changes into `hydra.run.dir` like `@hydra.main`.
"""
import copy
import datetime
import importlib
import os
import re
import sys
from typing import Any, Dict, List, Optional, Sequence

import yaml

# reference / third-party targets -> this package (used only if the original import fails)
TARGET_ALIASES = {
    "predict_pv_yield.models.conv3d.model.Model": "predict_pv_yield_amd.models.conv3d.model.Model",
    "predict_pv_yield.models.conv3d.model_sat_nwp.Model": "predict_pv_yield_amd.models.conv3d.model_sat_nwp.Model",
    "predict_pv_yield.models.conv3d.model_nwp.Model": "predict_pv_yield_amd.models.conv3d.model_nwp.Model",
    "predict_pv_yield.models.perceiver.perceiver.PerceiverModel": "predict_pv_yield_amd.models.perceiver.perceiver.PerceiverModel",
    "predict_pv_yield.models.perceiver.perceiver_nwp_sat.Model": "predict_pv_yield_amd.models.perceiver.perceiver_nwp_sat.Model",
    "predict_pv_yield.models.perceiver.perceiver_conv3d_nwp_sat.Model": "predict_pv_yield_amd.models.perceiver.perceiver_conv3d_nwp_sat.Model",
    "predict_pv_yield.models.baseline.last_value.Model": "predict_pv_yield_amd.models.baseline.last_value.Model",
    "predict_pv_yield.data.dataloader.NetCDFDataModule": "predict_pv_yield_amd.data.dataloader.NetCDFDataModule",
    "nowcasting_dataloader.datamodules.NetCDFDataModule": "predict_pv_yield_amd.data.dataloader.NetCDFDataModule",
    "pytorch_lightning.Trainer": "predict_pv_yield_amd.lightning.Trainer",
    "pytorch_lightning.callbacks.ModelCheckpoint": "predict_pv_yield_amd.lightning.ModelCheckpoint",
    "pytorch_lightning.callbacks.EarlyStopping": "predict_pv_yield_amd.lightning.EarlyStopping",
    "pytorch_lightning.loggers.csv_logs.CSVLogger": "predict_pv_yield_amd.lightning.CSVLogger",
}


class DictConfig(dict):
    """dict with attribute access (the subset of omegaconf.DictConfig the reference touches)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def __delattr__(self, k):
        del self[k]


def _wrap(v):
    if isinstance(v, dict):
        return DictConfig({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, list):
        return [_wrap(x) for x in v]
    return v


def _load_yaml(path: str) -> Dict[str, Any]:
    with open(path) as f:
        return yaml.safe_load(f) or {}


def _find(config_dir: str, group: str, name: str) -> str:
    name = name if name.endswith((".yaml", ".yml")) else name + ".yaml"
    path = os.path.join(config_dir, group, name)
    if not os.path.exists(path):
        raise FileNotFoundError(f"config group file not found: {group}/{name} under {config_dir}")
    return path


def _merge(dst: Dict[str, Any], src: Dict[str, Any]) -> Dict[str, Any]:
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = copy.deepcopy(v)
    return dst


def _set_path(cfg: Dict[str, Any], dotted: str, value: Any) -> None:
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        if not isinstance(node.get(k), dict):
            node[k] = {}
        node = node[k]
    node[keys[-1]] = value


def _del_path(cfg: Dict[str, Any], dotted: str) -> None:
    keys = dotted.split(".")
    node = cfg
    for k in keys[:-1]:
        node = node.get(k, {})
    node.pop(keys[-1], None)


def _get_path(cfg: Dict[str, Any], dotted: str) -> Any:
    node = cfg
    for k in dotted.split("."):
        node = node[k]
    return node


_INTERP = re.compile(r"\$\{([^${}]+)\}")


def _resolve_value(value: str, root: Dict[str, Any], cwd: str, depth: int = 0) -> Any:
    if depth > 20:
        raise ValueError(f"interpolation too deep: {value}")

    def repl(m):
        expr = m.group(1).strip()
        if expr.startswith("hydra:"):
            key = expr[len("hydra:"):]
            if key == "runtime.cwd":
                return cwd
            return str(_get_path(root.get("hydra", {}), key))
        if expr.startswith("oc.env:"):
            parts = expr[len("oc.env:"):].split(",", 1)
            val = os.environ.get(parts[0].strip(), parts[1].strip() if len(parts) > 1 else None)
            if val is None:
                raise KeyError(f"environment variable {parts[0]} is not set")
            return val
        if expr.startswith("now:"):
            return datetime.datetime.now().strftime(expr[len("now:"):])
        try:
            ref = _get_path(root, expr)
        except (KeyError, TypeError):
            return m.group(0).replace("${", "$\x00{")  # unresolvable here (e.g. hydra.job.num): keep literally
        if isinstance(ref, str):
            ref = _resolve_value(ref, root, cwd, depth + 1)
        return ref if isinstance(ref, str) else repr(ref)

    m = _INTERP.fullmatch(value.strip())
    if m and not m.group(1).startswith(("hydra:", "oc.env:", "now:")):
        try:
            ref = _get_path(root, m.group(1).strip())  # whole-value reference keeps its type
        except (KeyError, TypeError):
            return value
        return _resolve_value(ref, root, cwd, depth + 1) if isinstance(ref, str) else ref
    out = value
    while _INTERP.search(out):
        out = _INTERP.sub(repl, out)
    return out.replace("$\x00{", "${")


def resolve(cfg: Dict[str, Any], cwd: Optional[str] = None) -> Dict[str, Any]:
    cwd = cwd or os.getcwd()

    def walk(node):
        if isinstance(node, dict):
            for k in list(node):
                node[k] = walk(node[k])
            return node
        if isinstance(node, list):
            return [walk(x) for x in node]
        if isinstance(node, str) and "${" in node:
            return _resolve_value(node, cfg, cwd)
        return node

    return walk(cfg)


def _parse_scalar(text: str) -> Any:
    try:
        return yaml.safe_load(text)
    except yaml.YAMLError:
        return text


def compose(config_dir: str, config_name: str = "config", overrides: Sequence[str] = (), resolve_now: bool = True,
            cwd: Optional[str] = None) -> DictConfig:
    """hydra.compose(config_name, overrides) for the feature subset listed in the module docstring."""
    config_dir = os.path.abspath(config_dir)
    primary = _load_yaml(os.path.join(config_dir, config_name if config_name.endswith(".yaml") else config_name + ".yaml"))
    defaults = primary.pop("defaults", [])
    groups: Dict[str, Optional[str]] = {}
    order: List[str] = []
    for d in defaults:
        if isinstance(d, dict):
            (g, n), = d.items()
            g = g.replace("override ", "").lstrip("/")
            if g not in groups:
                order.append(g)
            groups[g] = n
    value_overrides, deletions = [], []
    for ov in overrides:
        if ov.startswith("~"):
            deletions.append(ov[1:])
            continue
        key, _, val = ov.partition("=")
        key = key.lstrip("+")
        if "." not in key and (key in groups or os.path.isdir(os.path.join(config_dir, key))):
            if key not in groups:
                order.append(key)
            groups[key] = None if val in ("null", "None", "") else val
        else:
            value_overrides.append((key, _parse_scalar(val)))

    # experiment overlays may re-select groups (`override /model: baseline.yaml`): read them first
    overlays = []
    for g in ("experiment", "hparams_search"):
        if groups.get(g):
            ov = _load_yaml(_find(config_dir, g, groups[g]))
            for d in ov.pop("defaults", []):
                if isinstance(d, dict):
                    (k, n), = d.items()
                    k = k.replace("override ", "").strip().lstrip("/")
                    # a command-line group choice wins over the experiment's
                    if not any(o.partition("=")[0] == k for o in overrides):
                        groups[k] = n
                        if k not in order:
                            order.append(k)
            overlays.append(ov)

    cfg: Dict[str, Any] = {}
    for g in order:
        if g in ("experiment", "hparams_search") or not groups.get(g):
            continue
        path = _find(config_dir, g, groups[g])
        body = _load_yaml(path)
        with open(path) as f:
            head = f.readline()
        if "@package _global_" in head:
            _merge(cfg, body)
        else:
            node = cfg
            for part in g.split("/"):
                node = node.setdefault(part, {})
            _merge(node, body)
    _merge(cfg, primary)
    for ov in overlays:
        _merge(cfg, ov)
    for k, v in value_overrides:
        _set_path(cfg, k, v)
    for k in deletions:
        _del_path(cfg, k)
    if resolve_now:
        resolve(cfg, cwd)
    return _wrap(cfg)


def _locate(path: str):
    module, _, attr = path.rpartition(".")
    return getattr(importlib.import_module(module), attr)


def get_class(target: str):
    try:
        return _locate(target)
    except (ImportError, AttributeError):
        if target in TARGET_ALIASES:
            return _locate(TARGET_ALIASES[target])
        raise


def instantiate(config: Dict[str, Any], *args, **kwargs):
    """hydra.utils.instantiate: `_target_(**config, **kwargs)`; nested `_target_` dicts are instantiated too."""
    if config is None:
        return None
    cfg = dict(config)
    target = cfg.pop("_target_")
    cfg.pop("_convert_", None)
    kwargs.pop("_convert_", None)
    recursive = cfg.pop("_recursive_", True)
    cfg.pop("_partial_", None)
    if recursive:
        for k, v in list(cfg.items()):
            if isinstance(v, dict) and "_target_" in v:
                cfg[k] = instantiate(v)
    cfg.update(kwargs)
    cls = get_class(target)
    plain = {k: (dict(v) if isinstance(v, DictConfig) else v) for k, v in cfg.items()}
    return cls(*args, **plain)


def to_yaml(cfg: Dict[str, Any]) -> str:
    def plain(v):
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, list):
            return [plain(x) for x in v]
        return v
    return yaml.safe_dump(plain(cfg), sort_keys=False)


def main(config_path: str, config_name: str = "config.yaml"):
    """@hydra.main replacement: composes from sys.argv overrides, chdirs into hydra.run.dir, calls fn(cfg)."""

    def deco(fn):
        def wrapper():
            caller_dir = os.path.dirname(os.path.abspath(sys.modules[fn.__module__].__file__))
            cfg_dir = config_path if os.path.isabs(config_path) else os.path.join(caller_dir, config_path)
            cwd = os.getcwd()
            cfg = compose(cfg_dir, config_name, sys.argv[1:], cwd=cwd)
            run_dir = cfg.get("hydra", {}).get("run", {}).get("dir")
            if run_dir:
                os.makedirs(run_dir, exist_ok=True)
                os.chdir(run_dir)
            try:
                return fn(cfg)
            finally:
                os.chdir(cwd)
        return wrapper
    return deco
