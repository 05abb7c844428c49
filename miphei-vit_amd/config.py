"""Composer for the reference's Hydra config tree and command line (Hydra / OmegaConf are not dependencies here).

The reference's entry points are ``@hydra.main(config_path="configs", config_name="config")``
(``/root/reference/run.py:16``), its tree is ``configs/config.yaml`` with a ``defaults:`` list
(``/root/reference/configs/config.yaml:1-5``), config groups ``data/ train/ model/`` and experiment overlays that are
``# @package _global_`` files carrying ``defaults: [override /train: cell]``
(``/root/reference/configs/default_configs/miphei-vit.yaml:1-3``), selected as ``+default_configs=miphei-vit``.
This module implements the subset of Hydra 1.3 semantics that tree and command line use:

* the defaults list of the primary config: ``_self_``, ``group: option``, ``optional group: option``, ``group: null``;
  when ``_self_`` is absent the primary config is merged last (Hydra >= 1.1);
* package of a group file: the group path (``train/cell.yaml`` lands under ``train``), or what a leading
  ``# @package <name>`` comment says (``_global_`` = root, ``_group_`` = the default, or a dotted path);
* ``defaults:`` inside group files: ``override /group: option`` re-selects an entry of the primary list (an unknown group is
  an error, as in Hydra), plain entries are composed recursively before the file itself;
* command line: ``group=option`` re-selects, ``+group=option`` appends a group, ``key=value`` changes an existing key (error if
  absent), ``+key=value`` adds a new key (error if present), ``++key=value`` does either, ``~key`` deletes;
  values are parsed as YAML; ``${a.b}`` interpolations of whole values or inside strings are resolved at the end.

The result is a nested mapping with attribute access (``cfg.model.encoder.encoder_name``).
"""
from __future__ import annotations

import os
import re

import yaml


class ConfigCompositionError(ValueError):
    pass


class Cfg(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def get_path(self, path, default=None):
        cur = self
        for key in path.split("."):
            if not isinstance(cur, dict) or key not in cur:
                return default
            cur = cur[key]
        return cur


def _wrap(o):
    if isinstance(o, dict):
        return Cfg({k: _wrap(v) for k, v in o.items()})
    if isinstance(o, list):
        return [_wrap(v) for v in o]
    return o


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v
    return dst


_PACKAGE = re.compile(r"^#\s*@package\s+(\S+)\s*$")


def _load(config_dir, rel):
    """(body without its defaults list, defaults list, package header or None) of ``<config_dir>/<rel>.yaml``."""
    path = os.path.join(config_dir, rel + ".yaml")
    if not os.path.isfile(path):
        group, _, name = rel.rpartition("/")
        have = sorted(f[:-5] for f in os.listdir(os.path.join(config_dir, group)) if f.endswith(".yaml")) \
            if os.path.isdir(os.path.join(config_dir, group)) else []
        raise ConfigCompositionError(f"Could not find '{rel}'; available options in '{group}': {have}")
    with open(path) as f:
        text = f.read()
    package = None
    for line in text.splitlines():
        if not line.strip():
            continue
        if not line.lstrip().startswith("#"):
            break
        m = _PACKAGE.match(line.strip())
        if m:
            package = m.group(1)
    body = yaml.safe_load(text) or {}
    if not isinstance(body, dict):
        raise ConfigCompositionError(f"{path}: top level must be a mapping")
    defaults = body.pop("defaults", [])
    return body, defaults, package


def _parse_default(entry, where):
    """-> ('self',) | ('group', group, option, is_override, optional)"""
    if entry == "_self_":
        return ("self",)
    if isinstance(entry, str):       # "- other_file": a config in the same group
        return ("group", None, entry, False, False)
    if isinstance(entry, dict) and len(entry) == 1:
        (k, v), = entry.items()
        words = k.split()
        override, optional = "override" in words[:-1], "optional" in words[:-1]
        return ("group", words[-1], v, override, optional)
    raise ConfigCompositionError(f"{where}: unsupported defaults entry {entry!r}")


def _place(root, package, body):
    if package in (None, "", "_global_"):
        return _merge(root, body)
    cur = root
    for part in package.split("."):
        nxt = cur.get(part)
        if not isinstance(nxt, dict):
            nxt = cur[part] = {}
        cur = nxt
    _merge(cur, body)
    return root


def _is_group(config_dir, key):
    return os.path.isdir(os.path.join(config_dir, key.strip("/")))


class _Composer:
    def __init__(self, config_dir, config_name):
        self.dir = config_dir
        self.primary, defaults, _ = _load(config_dir, config_name)
        self.entries = []            # primary defaults list: ["_self_"] | [group, option, optional]
        for d in defaults:
            p = _parse_default(d, config_name)
            if p[0] == "self":
                self.entries.append(["_self_"])
            else:
                _, group, option, override, optional = p
                if override:
                    raise ConfigCompositionError("'override' is not allowed in the primary defaults list")
                self.entries.append([group.strip("/"), option, optional])
        if ["_self_"] not in self.entries:
            self.entries.append(["_self_"])

    def select(self, group, option, must_exist=True):
        group = group.strip("/")
        for e in self.entries:
            if e[0] == group:
                e[1] = option
                return
        if must_exist:
            raise ConfigCompositionError(f"Could not override '{group}'. No match in the defaults list.")
        self.entries.append([group, option, False])

    def _collect_overrides(self, group, option, seen):
        """``override /g: o`` lines of a group file (and of the files its own defaults pull in) act on the primary list."""
        if option is None or (group, option) in seen:
            return
        seen.add((group, option))
        try:
            _, defaults, _ = _load(self.dir, f"{group}/{option}")
        except ConfigCompositionError:
            return                     # reported when the file is actually merged (unless the entry is optional)
        for d in defaults:
            p = _parse_default(d, f"{group}/{option}")
            if p[0] == "self":
                continue
            _, g, o, override, _ = p
            g = self._abs_group(group, g)
            if override:
                self.select(g, o)
            else:
                self._collect_overrides(g, o, seen)

    @staticmethod
    def _abs_group(parent, g):
        if g is None:
            return parent
        return g.strip("/") if g.startswith("/") else (f"{parent}/{g}" if parent else g)

    def _merge_group(self, root, group, option, optional, stack=()):
        if option is None:
            return
        if (group, option) in stack:
            raise ConfigCompositionError(f"defaults cycle through {group}/{option}")
        try:
            body, defaults, package = _load(self.dir, f"{group}/{option}")
        except ConfigCompositionError:
            if optional:
                return
            raise
        if package in (None, "_group_"):
            package = group.replace("/", ".")
        elif package.startswith("_group_."):
            package = group.replace("/", ".") + package[len("_group_"):]
        parsed = [_parse_default(d, f"{group}/{option}") for d in defaults]
        if not any(p[0] == "self" for p in parsed):
            parsed.append(("self",))
        for p in parsed:
            if p[0] == "self":
                _place(root, package, body)
            elif not p[3]:             # overrides were applied to the primary list up front
                self._merge_group(root, self._abs_group(group, p[1]), p[2], p[4], stack + ((group, option),))

    def compose(self):
        # overrides first, to a fixed point: a re-selected option may carry overrides of its own
        seen = set()
        for _ in range(8):
            before = [list(e) for e in self.entries]
            for e in list(self.entries):
                if e[0] != "_self_":
                    self._collect_overrides(e[0], e[1], seen)
            if before == self.entries:
                break
        root = {}
        for e in self.entries:
            if e[0] == "_self_":
                _merge(root, self.primary)
            else:
                self._merge_group(root, e[0], e[1], e[2])
        return root


def _walk(cfg, parts, create):
    cur = cfg
    for p in parts:
        nxt = cur.get(p) if isinstance(cur, dict) else None
        if not isinstance(nxt, dict):
            if not create:
                return None
            nxt = cur[p] = {}
        cur = nxt
    return cur


_INTERP = re.compile(r"\$\{([A-Za-z0-9_.]+)\}")


def _resolve(node, root, depth=0):
    if depth > 16:
        raise ConfigCompositionError("interpolation cycle")
    if isinstance(node, dict):
        for k in list(node):
            node[k] = _resolve(node[k], root, depth)
        return node
    if isinstance(node, list):
        return [_resolve(v, root, depth) for v in node]
    if isinstance(node, str) and "${" in node:
        def look(path):
            cur = root
            for p in path.split("."):
                if not isinstance(cur, dict) or p not in cur:
                    raise ConfigCompositionError(f"interpolation key '{path}' not found")
                cur = cur[p]
            return _resolve(cur, root, depth + 1)
        whole = _INTERP.fullmatch(node)
        if whole:
            return look(whole.group(1))
        return _INTERP.sub(lambda m: str(look(m.group(1))), node)
    return node


def compose(config_dir, overrides=(), config_name="config"):
    comp = _Composer(config_dir, config_name)
    sets = []
    for ov in overrides:
        if ov.startswith("~"):
            sets.append(("del", ov[1:].split("=", 1)[0], None))
            continue
        if "=" not in ov:
            raise ConfigCompositionError(f"override '{ov}': expected key=value")
        key, val = ov.split("=", 1)
        plus = len(key) - len(key.lstrip("+"))
        key = key.lstrip("+")
        if "." not in key and _is_group(config_dir, key) and plus < 2:
            # group selection: `train=structural` re-selects, `+default_configs=miphei-vit` appends
            comp.select(key, yaml.safe_load(val), must_exist=(plus == 0))
        else:
            sets.append((("set", "add", "force")[plus], key, yaml.safe_load(val)))
    cfg = comp.compose()
    for mode, key, val in sets:
        parts = key.split(".")
        parent = _walk(cfg, parts[:-1], create=(mode != "set" and mode != "del"))
        exists = isinstance(parent, dict) and parts[-1] in parent
        if mode == "del":
            if not exists:
                raise ConfigCompositionError(f"Could not delete from config. '{key}' does not exist.")
            del parent[parts[-1]]
            continue
        if mode == "set" and not exists:
            raise ConfigCompositionError(f"Could not override '{key}'.\nTo append to your config use +{key}={val}")
        if mode == "add" and exists:
            raise ConfigCompositionError(f"Could not append to config. An item is already at '{key}'. Either remove + "
                                         f"prefix: '{key}={val}'\nOr add a second + to add or override '{key}': '++{key}={val}'")
        parent[parts[-1]] = val
    return _wrap(_resolve(cfg, cfg))


def check_precision(value):
    """``train.precision`` as the reference hands it to ``pl.Trainer(precision=...)`` (/root/reference/src/train.py:205-207; the
    reference's shipped default is "16-mixed", /root/reference/configs/config.yaml:23).  The HIP training path has ONE arithmetic
    mode: bf16 MFMA operands, f32 accumulation, f32 master parameters and optimiser state -- Lightning's "bf16-mixed".  Returns
    "bf16-mixed"; "16-mixed" (fp16 autocast + GradScaler in the reference) is accepted with a warning because bf16 shares f32's
    exponent range and needs no loss scaler; anything else (true half / full / double precision) is refused instead of ignored."""
    import warnings
    v = str(value).strip().lower()
    if v in ("bf16-mixed", "bf16"):
        return "bf16-mixed"
    if v in ("16-mixed", "16"):
        warnings.warn('train.precision="16-mixed": the MI355X training path computes in bf16-mixed (bf16 MFMA operands, f32 '
                      'accumulation and master weights, no loss scaler); fp16 operands exist for evaluation only '
                      '(generator.eval().cuda().half())', UserWarning, stacklevel=2)
        return "bf16-mixed"
    raise NotImplementedError(f"train.precision={value!r}: the MI355X training path is bf16-mixed only "
                              '(accepted: "bf16-mixed", "16-mixed" with a warning)')
