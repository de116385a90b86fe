"""Multi-GPU tiling of the (k, m) task grid and the one exchange step of the path.

The reference fans (k, band, mode) tasks out to OS processes and pairs two Queues positionally on the way back
(Density_cylinder.py:1142-1168; latent mis-pairing race).  Here every rank owns a tile of the
(k, m) grid, produces (k, omega, m, resid, flag) records in one array, and the records are exchanged by a single
all-gather over RCCL (backend "nccl" on ROCm; "gloo" in the CPU tests): counts first, then the padded tables.
"""
import numpy as np


def tile_rows(n_rows, rank, world, strided=True):
    """Row indices (k or task indices) owned by `rank`. Strided tiling balances the k-dependent root density."""
    if strided:
        return np.arange(rank, n_rows, world)
    lo = (n_rows * rank) // world
    hi = (n_rows * (rank + 1)) // world
    return np.arange(lo, hi)


def tile_modes(m_values, rank, world):
    """Azimuthal orders owned by `rank` (round-robin)."""
    return [m for i, m in enumerate(m_values) if i % world == rank]


def pack_records(roots, m):
    """(n, 5) float64 records: k, omega, m, resid, flag -- accepted and rejected brackets alike."""
    import torch
    n = roots["w"].numel()
    rec = torch.empty((n, 5), dtype=torch.float64, device=roots["w"].device)
    rec[:, 0] = roots["k"]
    rec[:, 1] = roots["w"]
    rec[:, 2] = float(m)
    rec[:, 3] = roots["resid"]
    rec[:, 4] = roots["flag"].to(torch.float64)
    return rec


def gather_root_tables(roots, m, world=None, group=None):
    """All-gather the variable-length root tables of all ranks. Returns an (N_total, 5) tensor on every rank,
    rank-major (deterministic order)."""
    import torch
    import torch.distributed as dist
    rec = pack_records(roots, m)
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return rec
    home = rec.device
    if dist.get_backend(group) == "gloo" and rec.is_cuda:
        rec = rec.cpu()                 # rehearsal mode: gloo ranks sharing one GPU exchange through host memory
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=rec.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(max(counts), 1)
    padded = torch.zeros((nmax, 5), dtype=torch.float64, device=rec.device)
    padded[:rec.shape[0]] = rec
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded, group=group)
    return torch.cat([o[:c] for o, c in zip(out, counts)], dim=0).to(home)


# ---- fixed-capacity exchange: ONE collective per step, no host synchronisation -------------------------------------
# Record layout of the k-tiled grid search: (k, omega, m, resid, flag, global row).  Every rank sends a buffer of
# cap + 1 rows whose row 0 carries its bracket count, so counts and records travel in the same all-gather and nothing
# has to be read back on the host before the collective is enqueued (the two-phase exchange above needs the counts on
# the host first).  Ranks whose count exceeds `cap` are detected when the gathered buffer is merged.
N_FIELDS = 6


def pack_fixed(roots, count, m, rows_global, cap, ctx=None):
    """(cap + 1, 6) send buffer on the device of `roots`; rows_global[i] = global k-row of local row i (int64 tensor).
    `count`: the bracket count, an int or (GPU path) the one-element int32 CUDA tensor of find_roots_async -- then
    nothing is read back.  With a library context (GPU tensors) the buffer is filled by ONE kernel
    (es_root_table_pack[_async]) on the CONTEXT's stream: the buffers it touches are allocated under that stream, so
    the caching allocator cannot hand them out again while the kernel is pending, whatever torch's current stream is.
    The torch path is the same layout for CPU tensors (gloo tests)."""
    import torch
    dev = roots["w"].device
    if ctx is not None and roots["w"].is_cuda:
        import ctypes as C
        from . import _lib
        if not (isinstance(rows_global, torch.Tensor) and rows_global.device == dev):
            rows_global = torch.as_tensor(np.asarray(rows_global.cpu() if isinstance(rows_global, torch.Tensor) else rows_global),
                                          device=dev)
        with torch.cuda.stream(ctx.torch_stream):
            send = torch.empty((cap + 1, N_FIELDS), dtype=torch.float64, device=dev)
            rt = _lib.RootTable(roots["k"].data_ptr(), roots["w"].data_ptr(), roots["w_lo"].data_ptr(),
                                roots["w_hi"].data_ptr(), roots["resid"].data_ptr(), roots["row"].data_ptr(),
                                roots["flag"].data_ptr(), int(roots["w"].numel()))
            rg = rows_global.to(torch.int64).contiguous()
            if isinstance(count, torch.Tensor):
                assert count.is_cuda and count.numel() == 1 and count.element_size() == 4
                rc = ctx.lib.es_root_table_pack_async(ctx.handle, C.byref(rt), _lib.ptr(count), float(m),
                                                      C.c_void_p(rg.data_ptr()), int(cap), _lib.ptr(send))
            else:
                rc = ctx.lib.es_root_table_pack(ctx.handle, C.byref(rt), int(count), float(m),
                                                C.c_void_p(rg.data_ptr()), int(cap), _lib.ptr(send))
            _lib.check(ctx.handle, rc)
        for t_ in (send, rg):                       # consumers on other streams (the collective) order themselves by events
            t_.record_stream(ctx.torch_stream)
        return send
    send = torch.zeros((cap + 1, N_FIELDS), dtype=torch.float64, device=dev)
    count = int(count)
    n = min(count, cap, roots["w"].numel())
    send[0, 0] = float(count)
    if n > 0:
        send[1:n + 1, 0] = roots["k"][:n]
        send[1:n + 1, 1] = roots["w"][:n]
        send[1:n + 1, 2] = float(m)
        send[1:n + 1, 3] = roots["resid"][:n]
        send[1:n + 1, 4] = roots["flag"][:n].to(torch.float64)
        send[1:n + 1, 5] = rows_global[roots["row"][:n].long()].to(torch.float64)
    return send


def exchange_capacity(count, group=None, floor=64):
    """Capacity of the fixed-size exchange from the data: one all_reduce(MAX) of the ranks' bracket counts (warm-up,
    outside the timed region), then the next power of two >= 2 x that.  Round 2 sent a fixed 32768 records per rank --
    1.57 MB for about 820 records (39 KB) of a 512-row tile."""
    import torch
    import torch.distributed as dist
    c = int(count)
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
        t = torch.tensor([c], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        c = int(t.item())
    cap = floor
    while cap < 2 * c:
        cap *= 2
    return cap


def gather_fixed(send, world=None, group=None):
    """One all-gather of the fixed-size buffers -> (world, cap + 1, 6) on every rank (device of `send`; through host
    memory for gloo ranks that share a GPU)."""
    import torch
    import torch.distributed as dist
    if world is None:
        world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return send.unsqueeze(0)
    home = send.device
    if dist.get_backend(group) == "gloo" and send.is_cuda:
        send = send.cpu()
    out = torch.empty((world * send.shape[0], send.shape[1]), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send, group=group)          # concatenation along dim 0, rank-major
    return out.view(world, send.shape[0], send.shape[1]).to(home)


def merge_fixed(buf):
    """Gathered buffers -> (records sorted by (unit = field 2, global row, position within the rank's table), counts per
    rank).  With the k-rows of every unit (azimuthal order / mode) tiled across ranks this is exactly the single-GPU table
    order (units outer, rows next, omega inner).  Host side."""
    b = buf.detach().cpu().numpy()
    cap = b.shape[1] - 1
    counts = [int(round(b[r, 0, 0])) for r in range(b.shape[0])]
    if max(counts) > cap:
        raise OverflowError(f"root table of a rank has {max(counts)} records, exchange capacity {cap}")
    rec = np.concatenate([b[r, 1:1 + c] for r, c in enumerate(counts)], axis=0) if sum(counts) else np.zeros((0, N_FIELDS))
    order = np.lexsort((rec[:, 5], rec[:, 2]))            # stable: primary key unit, secondary key global row
    return rec[order], counts


def concat_fixed(sends):
    """Send buffers of several units (azimuthal orders / modes) of one rank -> ONE buffer for ONE all-gather per step:
    the (cap_u + 1)-row slots back to back, each with its own header row, nothing read back on the host."""
    import torch
    return sends[0] if len(sends) == 1 else torch.cat(sends, dim=0)


def merge_units(buf, caps):
    """merge_fixed for gathered concat_fixed buffers: buf is (world, sum(cap_u + 1), 6), caps the slot capacities in
    slot order.  Records sorted by (unit = field 2, global row, position), i.e. the single-GPU order of a run that
    handles the units one after the other; counts[rank][slot]."""
    b = buf.detach().cpu().numpy()
    assert b.shape[1] == sum(c + 1 for c in caps), (b.shape, caps)
    parts, counts = [], []
    for r in range(b.shape[0]):
        off, row_counts = 0, []
        for c in caps:
            n = int(round(b[r, off, 0]))
            if n > c:
                raise OverflowError(f"root table of a unit has {n} records, exchange capacity {c}")
            parts.append(b[r, off + 1:off + 1 + n])
            row_counts.append(n)
            off += c + 1
        counts.append(row_counts)
    rec = np.concatenate(parts, axis=0) if parts and sum(map(len, parts)) else np.zeros((0, N_FIELDS))
    order = np.lexsort((rec[:, 5], rec[:, 2]))
    return rec[order], counts


def gather_mode_results(local, group=None):
    """All-gather per-mode (omega, k) arrays of a k-tiled driver run: local = {"sausage": (w, k), "kink": (w, k)}
    (NumPy) -> the concatenation over ranks (rank-major) on every rank.  Same exchange pattern as
    gather_root_tables: counts, then padded records."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    out = {}
    for mode in sorted(local):
        w, k = local[mode]
        rec = torch.as_tensor(np.stack([np.asarray(w, dtype=np.float64), np.asarray(k, dtype=np.float64)], axis=1)
                              if len(w) else np.zeros((0, 2)), dtype=torch.float64, device=dev)
        n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
        counts = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(counts, n, group=group)
        counts = [int(c.item()) for c in counts]
        nmax = max(max(counts), 1)
        padded = torch.zeros((nmax, 2), dtype=torch.float64, device=dev)
        padded[:rec.shape[0]] = rec
        parts = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(parts, padded, group=group)
        allrec = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0).cpu().numpy()
        out[mode] = (allrec[:, 0].copy(), allrec[:, 1].copy())
    return out
