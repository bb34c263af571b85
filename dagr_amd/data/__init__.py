"""Minimal stand-ins for ``torch_geometric.data.Data`` / ``Batch`` (torch_geometric is not part of
this stack).  Attribute bag with the few behaviours the reference's hot path relies on:
``clone``, ``cuda``/``to``, ``num_graphs``, ``Batch.from_data_list`` with ``follow_batch``."""
import copy

import torch


class Data:
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        lazy = self.__dict__.get("_lazy") or {}
        return [k for k in self.__dict__.keys() if not k.startswith("_")] + [k for k in lazy if k not in self.__dict__]

    def __contains__(self, key):
        return key in self.__dict__ or key in (self.__dict__.get("_lazy") or {})

    # Attributes that are derived from others and cost launches to build (the reference-shaped ``edge_index`` /
    # ``edge_attr`` of a graph the kernels hold as CSR + offset codes) can be registered as recipes: they are built on
    # first access and behave like plain attributes from then on; assigning the attribute replaces the recipe.
    def set_lazy(self, name, fn):
        self.__dict__.pop(name, None)
        lazy = dict(self.__dict__.get("_lazy") or {})        # (never shared with the object this one was copied from)
        lazy[name] = fn
        self.__dict__["_lazy"] = lazy

    def is_lazy(self, name):
        return name not in self.__dict__ and name in (self.__dict__.get("_lazy") or {})

    def __getattr__(self, name):                              # reached only when the normal lookup fails
        lazy = self.__dict__.get("_lazy")
        if lazy is not None and name in lazy:
            value = lazy[name](self)
            self.__dict__[name] = value
            return value
        raise AttributeError(f"{type(self).__name__!r} object has no attribute {name!r}")

    def __setattr__(self, name, value):
        # ``edge_index`` assigned by the caller replaces the graph: the kernel-side forms derived from the OLD graph (CSR by
        # destination, integer pixel offsets, exact-offset cache) must not outlive it (set_lazy's contract: assigning the
        # attribute replaces the recipe).  The producers of those forms set them AFTER the graph they belong to.
        if name == "edge_index":
            for k in ("_dagr_csr", "_dagr_pixel_codes", "_dagr_exact"):
                self.__dict__.pop(k, None)
        object.__setattr__(self, name, value)

    def _apply(self, fn):
        out = copy.copy(self)
        for k in self.keys():
            v = getattr(self, k)
            if torch.is_tensor(v):
                out.__dict__[k] = fn(v)          # (the same graph on another device / in another dtype: caches stay)
        return out

    def to(self, device, non_blocking=False):
        return self._apply(lambda t: t.to(device, non_blocking=non_blocking))

    def cuda(self, non_blocking=False):
        return self.to("cuda", non_blocking=non_blocking)

    def cpu(self):
        return self.to("cpu")

    def clone(self):
        return self._apply(lambda t: t.clone())

    @property
    def num_nodes(self):
        for k in ("x", "pos"):
            v = getattr(self, k, None)
            if torch.is_tensor(v):
                return v.shape[0]
        return 0

    @property
    def num_graphs(self):
        if getattr(self, "_num_graphs", None) is not None:
            return self._num_graphs
        b = getattr(self, "batch", None)
        if b is None or b.numel() == 0:
            return 1
        return int(b.max().item()) + 1


class Batch(Data):
    @classmethod
    def from_data_list(cls, data_list, follow_batch=()):
        """Concatenate node-level tensors along dim 0 (PyG collation): x, pos, t, bbox, bbox0;
        ``image`` along dim 0; scalar attributes (width, height, time_window, ...) become 1-D tensors;
        strings become lists.  ``batch`` / ``<key>_batch`` hold the sample index."""
        out = cls()
        keys = data_list[0].keys()
        n_nodes = [d.num_nodes for d in data_list]
        for k in keys:
            vals = [getattr(d, k) for d in data_list]
            v0 = vals[0]
            if torch.is_tensor(v0) and v0.dim() > 0:
                setattr(out, k, torch.cat(vals, dim=0))
                if k in follow_batch:
                    setattr(out, k + "_batch", torch.cat([torch.full((v.shape[0],), i, dtype=torch.long)
                                                          for i, v in enumerate(vals)]))
            elif torch.is_tensor(v0) or isinstance(v0, (int, float)):
                setattr(out, k, torch.as_tensor([float(v) if isinstance(v, float) else int(v) for v in vals]))
            else:
                setattr(out, k, list(vals))
        out.batch = torch.cat([torch.full((n,), i, dtype=torch.long) for i, n in enumerate(n_nodes)])
        out._num_graphs = len(data_list)
        d0 = data_list[0]
        if all(hasattr(d0, k) for k in ("width", "height", "time_window")):
            # the sensor geometry as python ints: format_data / EV_TGN read it per batch, and the collated tensors move to
            # the device with the batch (a read-back there is a host synchronisation per scalar)
            out._geometry = tuple(int(getattr(d0, k)) for k in ("width", "height", "time_window"))
        return out


class DataLoader:
    """The slice of ``torch_geometric.data.DataLoader`` the scripts use (``run_test.py:48``, ``run_test_interframe.py:69``,
    ``train_ncaltech101.py:121-125``): sequential, sampler-ordered or shuffled batches collated with
    ``Batch.from_data_list``.  Two ways to split the work over ranks (``dagr_amd/parallel.py``), neither needs a collective:
    ``batches`` restricts the loader to this rank's batches (independent evaluation windows); ``shard=(rank, world)`` cuts
    EVERY batch into ``world`` equal slices (data-parallel training: all ranks walk the same seeded permutation, so the
    union of their slices is the global batch)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, sampler=None, follow_batch=(), drop_last=False,
                 num_workers=0, batches=None, shard=None, seed=0):
        self.dataset, self.batch_size = dataset, int(batch_size)
        self.order = list(sampler) if sampler is not None else list(range(len(dataset)))
        self.follow_batch, self.drop_last = tuple(follow_batch), bool(drop_last)
        n = len(self.order) // self.batch_size if drop_last else -(-len(self.order) // self.batch_size)
        self.batches = list(range(n)) if batches is None else [b for b in batches if b < n]
        self.shuffle, self.seed, self.epoch = bool(shuffle), int(seed), 0
        self.shard = shard
        if shard is not None and self.batch_size % shard[1]:
            raise ValueError(f"batch_size {self.batch_size} does not split over {shard[1]} ranks")

    def __len__(self):
        return len(self.batches)

    def __iter__(self):
        B = self.batch_size
        order = self.order
        if self.shuffle:
            g = torch.Generator().manual_seed(self.seed + self.epoch)
            order = [self.order[i] for i in torch.randperm(len(self.order), generator=g).tolist()]
            self.epoch += 1
        for k in self.batches:
            idx = order[k * B:(k + 1) * B]
            if self.shard is not None:
                rank, world = self.shard
                per = len(idx) // world
                idx = idx[rank * per:(rank + 1) * per]
            yield Batch.from_data_list([self._fetch(i) for i in idx], follow_batch=self.follow_batch)

    def image_ids(self, step):
        """Global positions, in the run, of the images of the ``step``-th batch THIS loader yields: batch k of the run
        holds positions [k*B, (k+1)*B); a ``batches`` loader yields whole batches of its own list, a ``shard`` loader the
        rank's slice of every batch.  The ids ``DetectionBuffer.update`` needs for one evaluation over all ranks."""
        B, k = self.batch_size, self.batches[step]
        ids = list(range(k * B, min((k + 1) * B, len(self.order))))
        if self.shard is not None:
            rank, world = self.shard
            per = len(ids) // world
            ids = ids[rank * per:(rank + 1) * per]
        return ids

    def _fetch(self, i):
        """Sample i of this epoch.  Shuffled (training) loaders draw the sample's random augmentations from a generator
        state derived from (seed, epoch, i), inside a forked RNG scope: the same sample is augmented the same way whatever
        the number of ranks or the order the ranks visit it in, and the caller's RNG stream is left untouched."""
        if not self.shuffle:
            return self.dataset[i]
        with torch.random.fork_rng(devices=[]):
            torch.manual_seed((self.seed * 1000003 + self.epoch) * 1000003 + int(i))
            return self.dataset[i]
