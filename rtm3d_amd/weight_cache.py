"""Cache of the BN-folded (fp32) and kernel-packed (fp16) weights of ONE state dict.

Folding BatchNorm into the convolutions in float64, composing the neck's 1x1 pairs and packing ~30 M weights into the
MFMA fragment order takes about 1.5 s of numpy per plan; none of it depends on the input shape except which kernel
variant a layer gets.  The cache keeps every result keyed by (layer, variant), is shared by all plans of a model,
is dropped by ``load_state_dict``, and can be written to / read from disk keyed by a digest of the state dict
(``RTM3D_WEIGHT_CACHE_DIR``), so that a serving process that re-loads the same checkpoint skips the work entirely.
"""
import hashlib
import os
import uuid
import zipfile

import numpy as np

# Bump when a packed layout changes.  The tag that goes into the digest, the file name and the file itself also carries the
# C ABI version and a hash of the packing code (plan.py), so that a cache file written by another build of the packers is
# never read back: same-sized blobs in a different fragment order would pass the runtime's size checks and convolve wrongly.
PACK_FORMAT_VERSION = 1


def format_tag():
    from . import _lib
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'plan.py'), 'rb') as f:
        src = hashlib.sha256(f.read()).hexdigest()[:12]
    return 'pack%d.abi%d.%s' % (PACK_FORMAT_VERSION, _lib.ABI_VERSION, src)


def state_dict_digest(sd, tag=None):
    """sha256 over the pack-format tag, key names, shapes, dtypes and raw bytes, in key order."""
    h = hashlib.sha256()
    h.update((format_tag() if tag is None else tag).encode())
    for k, v in sd.items():
        a = v.detach().cpu().numpy() if hasattr(v, 'detach') else np.asarray(v)
        h.update(k.encode()); h.update(str(a.dtype).encode()); h.update(str(a.shape).encode())
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


class WeightCache(object):
    def __init__(self, state_dict=None, directory=None):
        self.entries = {}          # key -> tuple of numpy arrays
        self.hits = self.misses = 0
        self.digest = None
        self.directory = directory if directory is not None else os.environ.get('RTM3D_WEIGHT_CACHE_DIR')
        self._dirty = False
        self.tag = format_tag()
        if state_dict is not None and self.directory:
            self.digest = state_dict_digest(state_dict, self.tag)
            self._load()

    def get(self, key, make):
        """entries[key], computed by ``make()`` (returning an array or a tuple of arrays) on first use."""
        v = self.entries.get(key)
        if v is None:
            self.misses += 1
            v = make()
            v = tuple(v) if isinstance(v, (tuple, list)) else (v,)
            self.entries[key] = v
            self._dirty = True
        else:
            self.hits += 1
        return v if len(v) > 1 else v[0]

    # ---- optional persistence
    def _path(self):
        return os.path.join(self.directory, 'rtm3d_weights_%s.npz' % self.digest[:32])

    def _load(self):
        """Read the file of this digest if there is one.  A file that cannot be read (truncated by a killed writer, another
        format tag, a missing part) is ignored: everything is recomputed and the next save() replaces it."""
        p = self._path()
        if not os.path.exists(p):
            return
        entries = {}
        try:
            with np.load(p, allow_pickle=False) as z:
                if str(z['__digest__']) != self.digest or str(z['__format__']) != self.tag:
                    return
                names = {}
                for name in z.files:
                    if name in ('__digest__', '__format__'):
                        continue
                    key, idx = name.rsplit('#', 1)
                    names.setdefault(key, {})[int(idx)] = z[name]
                for key, parts in names.items():
                    entries[key] = tuple(parts[i] for i in range(len(parts)))
        except (zipfile.BadZipFile, KeyError, ValueError, OSError, EOFError):
            return
        self.entries.update(entries)

    def save(self):
        """Write the cache next to its digest (no-op without a directory or when nothing new was computed).  The temporary
        file is private to this call (pid + uuid): the 8 ranks of one job loading the same checkpoint may all save at once."""
        if not self.directory or not self._dirty or self.digest is None:
            return None
        os.makedirs(self.directory, exist_ok=True)
        out = {'__digest__': np.array(self.digest), '__format__': np.array(self.tag)}
        for key, parts in self.entries.items():
            for i, a in enumerate(parts):
                out['%s#%d' % (key, i)] = a
        tmp = '%s.%d.%s.tmp.npz' % (self._path(), os.getpid(), uuid.uuid4().hex[:8])
        try:
            np.savez(tmp, **out)
            os.replace(tmp, self._path())
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
        self._dirty = False
        return self._path()
