"""Multi-GPU host layer: camera-slice partition of the normal equations + the two collectives the
C-ABI asks for (include/gsfm_rot.h, gsfm_rot_shard), implemented with torch.distributed.

One process per GPU.  Rank r owns the contiguous camera slice [r*P, (r+1)*P) (P = slice_width) and
receives every edge that touches an owned camera; per-camera sums (gradient, diagonal blocks,
A.p) are therefore complete on the owner and are exchanged with ONE in-place all-gather (no
reduction arithmetic, bitwise identical to the single-GPU sums); only scalars (cost) are
all-reduced.  Backend "nccl" (= RCCL over xGMI) operates directly on the device buffers on the
solver's stream; backend "gloo" (CPU tests, or several ranks sharing one GPU) stages through host.
"""
import ctypes as C
import os

import numpy as np

from . import _abi


class Partition(object):
    """Cameras -> ranks.  The C-ABI shards by contiguous slices of equal width P (gsfm_rot_shard.slice_width), so a partition is a
    relabelling of the cameras into the padded index space [0, world * P): rank r's cameras get the ids r*P, r*P + 1, ... and the
    rest of its slice is padding (cameras without edges, which the solver leaves untouched).  `new_id[old]` is that relabelling."""

    def __init__(self, n_cams, world, width, new_id, entries_per_rank):
        self.n_cams, self.world, self.width = int(n_cams), int(world), int(width)
        self.n_pad = self.world * self.width
        self.new_id = new_id
        self.entries_per_rank = entries_per_rank
        self.packed_components = False   # True: whole connected components per rank (pack_components); split_components of them had to be cut
        self.split_components = 0

    def scatter(self, per_camera, fill=0.0):
        """(n_cams, ...) in the caller's numbering -> (n_pad, ...) in the problem's."""
        a = np.asarray(per_camera)
        out = np.full((self.n_pad,) + a.shape[1:], fill, dtype=a.dtype)
        out[self.new_id] = a
        return out

    def gather(self, padded):
        return np.asarray(padded)[self.new_id]

    def relabel(self, cams):
        return self.new_id[np.asarray(cams, dtype=np.int64)].astype(np.uint32)


def locality_order(n_cams, edge_i, edge_j):
    """Camera order in which neighbours sit close together (the reverse Cuthill-McKee relabelling gsfm_rot_problem_create applies
    to unsharded problems, adopted under the same criterion), or the identity when the graph has no locality to recover (e.g. the
    uniformly random C5 graph).  Returns `order` with order[k] = the camera placed k-th."""
    lib = _abi.load_library()
    ei = np.ascontiguousarray(edge_i, dtype=np.uint32)
    ej = np.ascontiguousarray(edge_j, dtype=np.uint32)
    perm = np.empty(int(n_cams), dtype=np.uint32)
    adopted = lib.gsfm_rot_locality_order(int(n_cams), int(ei.size), ei.ctypes.data_as(C.POINTER(C.c_uint32)),
                                          ej.ctypes.data_as(C.POINTER(C.c_uint32)), perm.ctypes.data_as(C.POINTER(C.c_uint32)))
    if adopted < 0:
        raise ValueError("gsfm_rot_locality_order rejected the edge list")
    order = np.empty(int(n_cams), dtype=np.int64)
    order[perm.astype(np.int64)] = np.arange(int(n_cams))
    return order


def connected_components(n_cams, edge_i, edge_j):
    """Component label per camera (cameras without an edge are their own component), labels 0..n_comp-1 in order of first appearance."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components as cc
    ei, ej = np.asarray(edge_i, dtype=np.int64), np.asarray(edge_j, dtype=np.int64)
    n_comp, lab = cc(coo_matrix((np.ones(ei.size, dtype=np.int8), (ei, ej)), shape=(n_cams, n_cams)), directed=False)
    return int(n_comp), lab.astype(np.int64)


def pack_components(n_cams, edge_i, edge_j, world_size, order, deg, max_imbalance=1.25):
    """SURVEY 8(e), configuration C4 (several scenes batched as one disconnected graph): WHOLE connected components per rank, so that no edge is
    cut -- every rank holds exactly its own edges instead of its own plus the cut ones, and its mat-vec gathers stay inside its slice --
    balanced by directed entries with the longest-processing-time rule (pieces by decreasing size, each to the least loaded rank).  A scene
    heavier than a rank's fair share (Trafalgar among the 1DSfM scenes on 8 GPUs) is first cut into that many contiguous runs of its locality
    order, so only ITS edges can be cut.  Returns (per-rank camera lists, number of components that were split), or None when the graph is
    connected (one component with edges) or the packing still leaves the busiest rank more than `max_imbalance` x the mean."""
    n_comp, lab = connected_components(n_cams, edge_i, edge_j)
    load = np.bincount(lab, weights=deg, minlength=n_comp)
    with_edges = np.flatnonzero(load > 0)
    if with_edges.size < 2:
        return None
    pos = np.empty(n_cams, dtype=np.int64)
    pos[order] = np.arange(n_cams)
    by_comp = np.lexsort((pos, lab))                       # cameras grouped by component, locality order inside
    start = np.searchsorted(lab[by_comp], np.arange(n_comp + 1))
    target = load.sum() / world_size
    pieces, n_split = [], 0                               # (load, cameras)
    for c in with_edges:
        cams = by_comp[start[c]:start[c + 1]]
        k = int(max(1, np.ceil(load[c] / target - 0.25)))
        if k == 1:
            pieces.append((float(load[c]), cams))
            continue
        n_split += 1
        cs = np.cumsum(deg[cams])
        cuts = [0] + [int(np.searchsorted(cs, load[c] * t / k, side="left")) + 1 for t in range(1, k)] + [cams.size]
        for t in range(k):
            lo, hi = min(cuts[t], cams.size), min(max(cuts[t + 1], cuts[t]), cams.size)
            if hi > lo:
                pieces.append((float(deg[cams[lo:hi]].sum()), cams[lo:hi]))
    if len(pieces) < world_size:
        return None
    rank_load = np.zeros(world_size)
    members = [[] for _ in range(world_size)]
    for ld, cams in sorted(pieces, key=lambda p: -p[0]):
        r = int(np.argmin(rank_load))
        rank_load[r] += ld
        members[r].append(cams)
    if rank_load.max() > max_imbalance * rank_load.mean():
        return None
    # cameras without an edge: dealt out to even the camera counts (they cost nothing but a row of padding)
    iso = by_comp[np.isin(lab[by_comp], np.flatnonzero(load == 0))]
    counts = np.array([sum(c.size for c in m) for m in members])
    for chunk in np.array_split(iso, max(1, min(iso.size, 4 * world_size))) if iso.size else []:
        r = int(np.argmin(counts))
        members[r].append(chunk)
        counts[r] += chunk.size
    return [np.concatenate(m) if m else np.empty(0, dtype=np.int64) for m in members], n_split


def partition_cameras(n_cams, edge_i, edge_j, world_size, order=None, pack=True):
    """Contiguous slices of the locality ordering, cut so that every rank owns (nearly) the same number of directed entries
    (= block-CSR rows' worth of mat-vec work), every rank at least one camera.  In a spatially coherent view graph most
    neighbours of a rank's cameras then live on the same rank, so the gathers of the mat-vec stay inside its own slice; in a
    uniformly random graph every balanced partition cuts (world-1)/world of the edges and only the balance matters.
    A DISCONNECTED graph with enough components is packed instead -- whole components per rank, no edge cut (pack_components)."""
    n_cams, world_size = int(n_cams), int(world_size)
    if world_size < 1 or n_cams < world_size:
        raise ValueError("need at least one camera per rank (%d cameras, %d ranks)" % (n_cams, world_size))
    if order is None:
        order = locality_order(n_cams, edge_i, edge_j)
    order = np.asarray(order, dtype=np.int64)
    deg = np.bincount(np.asarray(edge_i, dtype=np.int64), minlength=n_cams) + np.bincount(np.asarray(edge_j, dtype=np.int64), minlength=n_cams)
    if pack and world_size > 1:
        packed = pack_components(n_cams, edge_i, edge_j, world_size, order, deg)
        if packed is not None and all(m.size > 0 for m in packed[0]):
            members, n_split = packed
            width = max(m.size for m in members)
            new_id = np.empty(n_cams, dtype=np.int64)
            for r, m in enumerate(members):
                new_id[m] = r * width + np.arange(m.size)
            part = Partition(n_cams, world_size, width, new_id, [int(deg[m].sum()) for m in members])
            part.packed_components, part.split_components = True, n_split
            assert np.unique(new_id).size == n_cams
            return part
    csum = np.cumsum(deg[order])
    total = int(csum[-1]) if n_cams else 0
    cuts = [0]
    for r in range(1, world_size):
        c = int(np.searchsorted(csum, r * total / world_size, side="left")) + 1 if total else (r * n_cams) // world_size
        c = max(c, cuts[-1] + 1)                           # at least one camera per rank ...
        c = min(c, n_cams - (world_size - r))              # ... also for the ranks still to come
        cuts.append(c)
    cuts.append(n_cams)
    width = max(cuts[r + 1] - cuts[r] for r in range(world_size))
    new_id = np.empty(n_cams, dtype=np.int64)
    entries = []
    for r in range(world_size):
        members = order[cuts[r]:cuts[r + 1]]
        new_id[members] = r * width + np.arange(members.size)
        entries.append(int(deg[members].sum()))
    part = Partition(n_cams, world_size, width, new_id, entries)
    assert np.unique(new_id).size == n_cams and new_id.min() >= 0 and new_id.max() < part.n_pad
    return part


def local_edge_mask(part, edge_i, edge_j, rank):
    """Edges a rank must hold: those touching one of its cameras (edge ids in the partition's numbering)."""
    lo, hi = rank * part.width, (rank + 1) * part.width
    ei = np.asarray(edge_i, dtype=np.int64)
    ej = np.asarray(edge_j, dtype=np.int64)
    return ((ei >= lo) & (ei < hi)) | ((ej >= lo) & (ej < hi))


def cost_owner(part, edge_i, edge_j):
    """The rank that counts an edge's rho in the cost (rule of gsfm_rot_problem_create)."""
    ei = np.asarray(edge_i, dtype=np.int64)
    ej = np.asarray(edge_j, dtype=np.int64)
    c = np.where(((ei + ej) & 1) == 0, ei, ej)
    return c // part.width


class _DevView(object):
    """Minimal __cuda_array_interface__ carrier so torch can alias a raw device pointer."""

    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


class TorchComm(object):
    """Owns the ctypes callbacks handed to the library through gsfm_rot_shard."""

    def __init__(self, width, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.group = torch, dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        self.P = int(width)
        self._views = {}
        self._host = {}
        self._ext = {}
        self.n_all_gather = 0
        self.n_all_reduce = 0
        self._ag = _abi.ALL_GATHER_FN(self._all_gather)
        self._ar = _abi.ALL_REDUCE_FN(self._all_reduce)
        self.shard = _abi.Shard(rank=self.rank, world_size=self.world, slice_width=self.P, flags=0, ctx=None,
                                all_gather=self._ag, all_reduce_sum=self._ar)

    def stream_handle(self):
        """The solver keeps its own stream; the callbacks order torch's collectives against the stream handed to them."""
        return None

    def _on_solver_stream(self, stream):
        """Context in which torch's current stream IS the solver's stream: ProcessGroupNCCL makes its communication stream wait
        for the current stream before the collective and the current stream wait for the collective after it."""
        key = int(stream) if stream else 0
        ext = self._ext.get(key)
        if ext is None:
            ext = self.torch.cuda.ExternalStream(key) if key else self.torch.cuda.current_stream()
            self._ext[key] = ext
        return self.torch.cuda.stream(ext)

    def _view(self, ptr, count):
        key = (ptr, count)
        t = self._views.get(key)
        if t is None:
            t = self.torch.as_tensor(_DevView(ptr, count), device="cuda")
            self._views[key] = t
        return t

    def _host_buf(self, count):
        t = self._host.get(count)
        if t is None:
            t = self.torch.empty(count, dtype=self.torch.float64).pin_memory()
            self._host[count] = t
        return t

    def _sync_stream(self, stream):
        if stream:
            self.torch.cuda.ExternalStream(int(stream)).synchronize()
        else:
            self.torch.cuda.synchronize()

    def _all_gather(self, _ctx, buf, count, stream):
        try:
            self.n_all_gather += 1
            total = count * self.world
            full = self._view(buf, total)
            mine = full[self.rank * count:(self.rank + 1) * count]
            if self.backend == "nccl":
                with self._on_solver_stream(stream):
                    self.dist.all_gather_into_tensor(full, mine, group=self.group)
            else:
                self._sync_stream(stream)
                h_in = mine.cpu()
                h_out = self._host_buf(total)
                self.dist.all_gather_into_tensor(h_out, h_in, group=self.group)
                full.copy_(h_out)
                self._sync_stream(None)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            import sys
            print("gsfm all_gather callback failed: %r" % (e,), file=sys.stderr)
            return 1

    def _all_reduce(self, _ctx, buf, count, stream):
        try:
            self.n_all_reduce += 1
            t = self._view(buf, count)
            if self.backend == "nccl":
                with self._on_solver_stream(stream):
                    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
            else:
                self._sync_stream(stream)
                h = t.cpu()
                self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
                t.copy_(h)
                self._sync_stream(None)
            return 0
        except Exception as e:
            import sys
            print("gsfm all_reduce callback failed: %r" % (e,), file=sys.stderr)
            return 1


class NativeComm(object):
    """The same two collectives issued by C++ (globalsfmpy_amd/csrc/gsfm_rccl.cpp) straight into RCCL on the
    solver's own stream.  torch.distributed is used once, to hand the ncclUniqueId to every rank."""

    def __init__(self, width, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.P = int(width)
        # Every step that can fail is agreed on by all ranks before the next collective: a rank that raised alone (a missing
        # libgsfm_rccl.so included) would leave the others blocked inside ncclCommInitRank or inside the broadcast of the unique id.
        def all_ok(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(t.item())

        here = os.path.dirname(os.path.abspath(__file__))
        lib, why = None, ""
        try:
            lib = C.CDLL(os.path.join(here, "libgsfm_rccl.so"))
            lib.gsfm_rccl_last_error.restype = C.c_char_p
            lib.gsfm_rccl_create.restype = C.c_void_p
            lib.gsfm_rccl_create.argtypes = [C.c_char_p, C.c_int, C.c_int]
            lib.gsfm_rccl_destroy.argtypes = [C.c_void_p]
            lib.gsfm_rccl_init.argtypes = [C.c_char_p]
            bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            if lib.gsfm_rccl_init(bundled.encode() if os.path.exists(bundled) else None) != 0:
                why, lib = lib.gsfm_rccl_last_error().decode(), None
        except OSError as e:
            why, lib = str(e), None
        if not all_ok(lib is not None):
            raise RuntimeError("RCCL not available on every rank%s" % (": " + why if why else " (this rank is fine)"))
        ident = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(128)
            if lib.gsfm_rccl_unique_id(buf) == 0:
                ident[0] = buf.raw
        # (`src` is a GLOBAL rank: group rank 0 of a sub-group need not be global rank 0)
        dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if ident[0] is None:
            raise RuntimeError("ncclGetUniqueId failed on rank 0: %s" % lib.gsfm_rccl_last_error().decode())
        torch.cuda.synchronize()
        self._ctx = lib.gsfm_rccl_create(ident[0], self.rank, self.world)
        if not all_ok(bool(self._ctx)):
            if self._ctx:
                lib.gsfm_rccl_destroy(self._ctx)
                self._ctx = None
            raise RuntimeError("ncclCommInitRank failed on some rank: %s" % lib.gsfm_rccl_last_error().decode())
        self._lib = lib
        self.backend = "rccl-native"
        self.n_all_gather = self.n_all_reduce = -1  # not counted on this path
        ag = C.cast(lib.gsfm_rccl_all_gather, C.c_void_p).value
        ar = C.cast(lib.gsfm_rccl_all_reduce_sum, C.c_void_p).value
        self.shard = _abi.Shard(rank=self.rank, world_size=self.world, slice_width=self.P, flags=_abi.SHARD_CAPTURABLE, ctx=self._ctx,
                                all_gather=_abi.ALL_GATHER_FN(ag), all_reduce_sum=_abi.ALL_REDUCE_FN(ar))

    def stream_handle(self):
        return None  # the problem's own non-blocking stream

    def close(self):
        if self._ctx:
            self._lib.gsfm_rccl_destroy(self._ctx)
            self._ctx = None


class PeerComm(object):
    """The two collectives as PEER STORES (csrc/gsfm_peer.hip): every rank's slice is written straight into a mailbox of every other rank
    through hipIpcMemHandle-mapped pointers (over xGMI on a multi-GPU node) and a flag word tells the owner it has arrived -- two small
    kernels on the solver's stream per collective, no collective-library launch inside the PCG loop, capturable into the PCG hipGraphs.
    Calls that do not fit the mailbox (sized for the per-camera exchanges of the solve) go to `fallback` (NativeComm / TorchComm).
    torch.distributed is used once, to exchange the IPC handles."""

    def __init__(self, width, fallback, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist, self.fallback = torch, dist, fallback
        self.rank, self.world, self.P = dist.get_rank(group), dist.get_world_size(group), int(width)

        def all_ok(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda" if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(t.item())

        here = os.path.dirname(os.path.abspath(__file__))
        lib, ctx, why = None, None, ""
        try:
            lib = C.CDLL(os.path.join(here, "libgsfm_peer.so"))
            lib.gsfm_peer_last_error.restype = C.c_char_p
            lib.gsfm_peer_create.restype = C.c_void_p
            lib.gsfm_peer_create.argtypes = [C.c_int, C.c_int, C.c_size_t, C.c_char_p]
            lib.gsfm_peer_connect.argtypes = [C.c_void_p, C.c_char_p]
            lib.gsfm_peer_set_fallback.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
            lib.gsfm_peer_destroy.argtypes = [C.c_void_p]
            lib.gsfm_peer_error.argtypes = [C.c_void_p]
            lib.gsfm_peer_error_take.argtypes = [C.c_void_p]
            lib.gsfm_peer_inject_error.argtypes = [C.c_void_p]
            lib.gsfm_peer_calls.argtypes = [C.c_void_p, C.c_int]
            lib.gsfm_peer_calls.restype = C.c_long
            nb = lib.gsfm_peer_handle_bytes()
            handle = C.create_string_buffer(nb)
            # gD slices (9 per camera) are the widest per-camera exchange; the PCG slot is 3 per camera + at most 8192 partial dot products
            cap = 9 * self.P + 16384
            # SHARD_CAPTURABLE below is a promise about the calls INSIDE a PCG chunk: the all-gather of the A.p slices (3 per camera) with, in the
            # single-reduction recurrence, the rank's partial dot products in its tail (8 per block of 256 cameras: problem_create.hpp, w_tail).
            # They fit the mailbox by construction -- checked here, so that no in-chunk call can reach a host-staged fallback under capture
            # (round-4 advisor); everything wider (the coarse matrix, once per LM step) is issued outside the chunks.
            assert 3 * self.P + 8 * ((self.P + 255) // 256) <= cap
            ctx = lib.gsfm_peer_create(self.rank, self.world, cap, handle)
            if not ctx:
                why = lib.gsfm_peer_last_error().decode()
        except OSError as e:
            why = str(e)
        if not all_ok(bool(ctx)):
            if ctx:
                lib.gsfm_peer_destroy(ctx)
            raise RuntimeError("peer exchange unavailable on some rank%s" % (": " + why if why else ""))
        handles = [None] * self.world
        dist.all_gather_object(handles, handle.raw, group=group)
        rc = lib.gsfm_peer_connect(ctx, b"".join(handles))
        if not all_ok(rc == 0):
            why = lib.gsfm_peer_last_error().decode() if rc else ""
            lib.gsfm_peer_destroy(ctx)
            raise RuntimeError("peer exchange: a mailbox could not be mapped%s" % (": " + why if why else " (on another rank)"))
        fs = fallback.shard
        lib.gsfm_peer_set_fallback(ctx, fs.ctx, C.cast(fs.all_gather, C.c_void_p), C.cast(fs.all_reduce_sum, C.c_void_p))
        self._lib, self._ctx = lib, ctx
        self.backend = "peer-store+" + fallback.backend
        self.n_all_gather = self.n_all_reduce = -1
        ag = C.cast(lib.gsfm_peer_all_gather, C.c_void_p).value
        ar = C.cast(lib.gsfm_peer_all_reduce_sum, C.c_void_p).value
        # capturable: every collective inside a PCG chunk fits the mailbox, i.e. is two plain kernel launches on the solver's stream
        self.shard = _abi.Shard(rank=self.rank, world_size=self.world, slice_width=self.P, flags=_abi.SHARD_CAPTURABLE, ctx=ctx,
                                all_gather=_abi.ALL_GATHER_FN(ag), all_reduce_sum=_abi.ALL_REDUCE_FN(ar))

    def stream_handle(self):
        return None

    def calls(self):
        """(collectives served by peer stores, collectives handed to the fallback)"""
        return int(self._lib.gsfm_peer_calls(self._ctx, 0)), int(self._lib.gsfm_peer_calls(self._ctx, 1))

    def error(self):
        """True if a wait for a peer's flag ever ran into its bound (that solve's result is then invalid; every later call of this
        communicator goes to the fallback collectives)."""
        return bool(self._lib.gsfm_peer_error(self._ctx))

    def take_error(self):
        """True exactly once: a wait timed out and no collective call has returned the failure to the solver yet (the time-out happened in the
        last calls of a solve, or inside a replayed hipGraph).  RotationProblem checks it after every library call and raises."""
        return bool(self._lib.gsfm_peer_error_take(self._ctx))

    def close(self):
        if self._ctx:
            self.torch.cuda.synchronize()
            self.dist.barrier()          # nobody unmaps a mailbox a peer may still be storing into
            self._lib.gsfm_peer_destroy(self._ctx)
            self._ctx = None
        if hasattr(self.fallback, "close"):
            self.fallback.close()


def make_comm(width, prefer_native=True, group=None, exchange=None):
    """The two collectives for slices of `width` cameras (Partition.width): NativeComm when RCCL can be driven from C++ (backend nccl),
    else the torch.distributed callbacks."""
    import torch.distributed as dist
    exchange = exchange or os.environ.get("GSFM_EXCHANGE", "collective")   # "peer": peer-store exchange (PeerComm) over the comm chosen below
    comm = None
    if prefer_native and dist.get_backend(group) == "nccl" and os.environ.get("GSFM_NO_NATIVE_RCCL") is None:
        try:
            comm = NativeComm(width, group)
        except Exception as e:  # noqa: BLE001
            import sys
            print("gsfm: native RCCL unavailable (%r); falling back to torch.distributed collectives" % (e,), file=sys.stderr)
    if comm is None:
        comm = TorchComm(width, group)
    if exchange == "peer":
        try:
            return PeerComm(width, comm, group)
        except Exception as e:  # noqa: BLE001  (every rank raises together: the failure is agreed on inside PeerComm)
            import sys
            print("gsfm: peer-store exchange unavailable (%r); using %s collectives" % (e, comm.backend), file=sys.stderr)
    return comm


def make_sharded_problem(graph, error_type, loss=None, prefer_native=True, group=None, part=None, exchange=None):
    """graph: dict from synth.make_graph (global, identical on every rank).  Partitions the cameras, builds this rank's share of the
    problem and returns (problem, partition); per-camera arrays enter and leave through partition.scatter / .gather."""
    import torch.distributed as dist
    from .solver import RotationProblem
    n = graph["n_cams"]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if part is None:
        part = partition_cameras(n, graph["edge_i"], graph["edge_j"], world)
    comm = make_comm(part.width, prefer_native, group, exchange)
    # (whether the GLOBAL graph is disconnected -- it decides the PCG tolerance -- is worked out inside gsfm_rot_problem_create from all
    # ranks' edges since round 3; the SHARD_DISCONNECTED flag is no longer needed here)
    ei, ej = part.relabel(graph["edge_i"]), part.relabel(graph["edge_j"])
    m = local_edge_mask(part, ei, ej, rank)
    cov6 = graph["cov6"][m] if graph.get("cov6") is not None else None
    inl = graph["inlier_weight"][m] if graph.get("inlier_weight") is not None else None
    prob = RotationProblem(part.n_pad, ei[m], ej[m], graph["rel_aa"][m], error_type, cov6=cov6, inlier_weight=inl,
                           shard=comm.shard, stream=comm.stream_handle())
    prob._comm = comm
    if loss is not None:
        prob.set_loss(loss)
    return prob, part
