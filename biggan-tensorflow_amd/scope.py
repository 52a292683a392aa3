"""``tf.variable_scope`` / ``tf.get_variable`` emulation for eager execution on device tensors.

The reference's ops create and look up parameters implicitly by scope path (ops.py:50,88,97,...);
the names are also the train-variable partition ('generator' / 'discriminator' substrings,
BigGAN.py:915-917) and the weight-interchange keys.  This module reproduces those semantics:

* nested scopes join with '/', ``reuse`` is inherited;
* ``variable_scope(None, default_name=...)`` picks ``name``, ``name_1``, ... unique within the
  parent (ops.py:533), the counters of sub-scopes being cleared when a scope closes, so that every
  re-execution of a model function resolves the same names (TF's ``close_variable_subscopes``);
* trainable variables of one network live in ONE flat fp32 arena (with matching flat arenas for
  gradients, Adam moments and EMA shadows) so the optimiser is a single kernel and the
  data-parallel gradient exchange a handful of large RCCL calls (``VariableStore.pack``).
"""
from collections import OrderedDict
from contextlib import contextmanager

import numpy as np
import torch


# ------------------------------------------------------------------------------------------
# initialisers (ops.py:13, 97, 534, 722)
# ------------------------------------------------------------------------------------------
class truncated_normal_initializer:
    def __init__(self, mean=0.0, stddev=1.0):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, rng):
        a = rng.standard_normal(shape)
        bad = np.abs(a) > 2.0
        while bad.any():                      # TF re-draws values beyond two standard deviations
            a[bad] = rng.standard_normal(int(bad.sum()))
            bad = np.abs(a) > 2.0
        return (a * self.stddev + self.mean).astype(np.float32)


class random_normal_initializer:
    def __init__(self, mean=0.0, stddev=1.0):
        self.mean, self.stddev = mean, stddev

    def __call__(self, shape, rng):
        return (rng.standard_normal(shape) * self.stddev + self.mean).astype(np.float32)


class constant_initializer:
    def __init__(self, value=0.0):
        self.value = value

    def __call__(self, shape, rng):
        return np.full(shape, self.value, dtype=np.float32)


# ------------------------------------------------------------------------------------------
# variable store
# ------------------------------------------------------------------------------------------
class Arena:
    """Flat fp32 buffers for one network's trainables: params, grads, Adam m / v, EMA shadow."""

    ALIGN = 64   # floats (256 B): every view is 16-byte aligned for the vector kernels

    def __init__(self, names, tensors, device, with_ema):
        self.names = list(names)
        self.offsets = OrderedDict()
        off = 0
        for n, t in zip(names, tensors):
            self.offsets[n] = (off, t.numel(), tuple(t.shape))
            off += (t.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.size = max(off, self.ALIGN)
        self.params = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.grads = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.m = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.v = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.ema = torch.zeros(self.size, dtype=torch.float32, device=device) if with_ema else None
        self.step = 0

    def view(self, buf, name):
        off, n, shape = self.offsets[name]
        return buf.narrow(0, off, n).view(shape)


class VariableStore:
    def __init__(self, device="cuda", seed=42):
        self.device = torch.device(device)
        self.rng = np.random.default_rng(seed)
        self.vars = OrderedDict()            # name -> tensor (leaf)
        self.trainable = OrderedDict()       # name -> bool
        self.regularizers = OrderedDict()    # name -> callable(w) -> scalar loss   (tf regulariser attached at creation)
        self.arenas = {}                     # 'generator' / 'discriminator' -> Arena
        self._stack = []                     # [(full_name, reuse)]
        self._counts = {}                    # scope path -> times opened (for default_name uniquifying)
        self._kids = {}                      # scope path -> paths of the scopes opened directly inside it
        self.frozen = False
        self.sn_pairs = OrderedDict()        # spectrally-normalised weight name -> name of its u vector
        self.reg_shapes = OrderedDict()      # regularised kernel name -> shape (tf regularisation-loss collection)

    def register_sn(self, w_name, u_name):
        self.sn_pairs.setdefault(w_name, u_name)

    # ---- scope handling -------------------------------------------------------------------
    @property
    def scope_name(self):
        return self._stack[-1][0] if self._stack else ""

    @property
    def reuse(self):
        return self._stack[-1][1] if self._stack else False

    def _unique(self, prefix):
        cur = self.scope_name
        name = cur + "/" + prefix if cur else prefix
        if self._counts.get(name, 0) == 0:
            return prefix
        idx = 1
        while self._counts.get(name + "_%d" % idx, 0) > 0:
            idx += 1
        return prefix + "_%d" % idx

    @contextmanager
    def variable_scope(self, name_or_scope=None, default_name=None, reuse=None):
        if name_or_scope is None:
            if default_name is None:
                raise ValueError("variable_scope needs a name or a default_name")
            name_or_scope = self._unique(default_name)
        cur = self.scope_name
        full = cur + "/" + name_or_scope if cur else name_or_scope
        self._counts[full] = self._counts.get(full, 0) + 1
        self._kids.setdefault(cur, set()).add(full)
        inherited = self.reuse if reuse is None else reuse
        self._stack.append((full, inherited))
        try:
            yield full
        finally:
            self._stack.pop()
            # TF close_variable_subscopes: the default-name counters of everything below `full` start over.  (Walks the
            # sub-scope tree; a scan of every counter with startswith() cost 2.5 ms of host time per iteration.)
            todo = [full]
            while todo:
                for k in self._kids.get(todo.pop(), ()):
                    self._counts[k] = 0
                    todo.append(k)

    # ---- variables ------------------------------------------------------------------------
    def get_variable(self, name, shape, initializer=None, trainable=True, regularizer=None):
        full = self.scope_name + "/" + name if self.scope_name else name
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else [shape]))
        if full in self.vars:
            v = self.vars[full]
            if tuple(v.shape) != shape:
                raise ValueError("Trying to share variable %s, but specified shape %s and found shape %s."
                                 % (full, shape, tuple(v.shape)))
            return v
        if self.frozen:
            raise ValueError("Variable %s does not exist (store is frozen after pack())" % full)
        init = initializer if initializer is not None else truncated_normal_initializer(0.0, 0.02)
        t = torch.from_numpy(init(shape, self.rng)).to(self.device)
        t.requires_grad_(bool(trainable))
        t.bg_name = full
        t.bg_grad = None          # gradient slot (arena view), set by pack()
        t.bg_touched = False
        self.vars[full] = t
        self.trainable[full] = bool(trainable)
        if regularizer is not None and trainable:
            self.regularizers[full] = regularizer
        return t

    def trainable_variables(self, substring=None):
        return OrderedDict((k, v) for k, v in self.vars.items()
                           if self.trainable[k] and (substring is None or substring in k))

    # ---- flat arenas ----------------------------------------------------------------------
    def pack(self, groups=("generator", "discriminator"), ema_groups=("generator",)):
        """Move every trainable of each group into one flat arena (values preserved) and freeze the
        store.  Afterwards ``var.bg_grad`` is the variable's slice of the flat gradient buffer."""
        for g in groups:
            tv = self.trainable_variables(g)
            if not tv:
                continue
            arena = Arena(list(tv.keys()), list(tv.values()), self.device, g in ema_groups)
            for name, old in tv.items():
                view = arena.view(arena.params, name)
                with torch.no_grad():
                    view.copy_(old)
                view.requires_grad_(True)
                view.bg_name = name
                view.bg_grad = arena.view(arena.grads, name)
                view.bg_touched = False
                self.vars[name] = view
            if arena.ema is not None:
                arena.ema.copy_(arena.params)      # shadows start at the initial values
            self.arenas[g] = arena
        self.frozen = True

    def begin_backward(self, group):
        """Mark every gradient slot of the group as not-yet-written for this step."""
        for name in self.arenas[group].names:
            self.vars[name].bg_touched = False

    def zero_untouched(self, group, lo=0, hi=None):
        """Zero the gradient slots no kernel wrote this step (optionally only the slots inside [lo, hi) of the arena)."""
        arena = self.arenas[group]
        for name in arena.names:
            off = arena.offsets[name][0]
            if off < lo or (hi is not None and off >= hi):
                continue
            v = self.vars[name]
            if not v.bg_touched:
                v.bg_grad.zero_()

    # ---- import / export (checkpoint & parity interchange keyed by TF variable names) ---------
    def load_arrays(self, arrays, strict=True, reset_ema=True):
        with torch.no_grad():
            for k, a in arrays.items():
                if k not in self.vars:
                    if strict:
                        raise KeyError(k)
                    continue
                self.vars[k].copy_(torch.as_tensor(np.asarray(a), dtype=torch.float32).to(self.device))
        if reset_ema:
            for arena in self.arenas.values():
                if arena.ema is not None:
                    arena.ema.copy_(arena.params)

    def export_arrays(self):
        return OrderedDict((k, v.detach().cpu().numpy().copy()) for k, v in self.vars.items())


# the default store used by the ops (one model per process, like one TF graph)
_default = None


def default_store():
    global _default
    if _default is None:
        _default = VariableStore()
    return _default


def set_default_store(store):
    global _default
    _default = store
    return store


def variable_scope(name_or_scope=None, default_name=None, reuse=None):
    return default_store().variable_scope(name_or_scope, default_name, reuse)


def get_variable(name, shape, initializer=None, trainable=True, regularizer=None, dtype=None):
    return default_store().get_variable(name, shape, initializer, trainable, regularizer)


def get_variable_scope_name():
    return default_store().scope_name
