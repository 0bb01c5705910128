"""Delay-grid sharding over the GPUs of one node: one process per GPU (torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests).

The reference parallelises the same loop with Distributed.pmap over candidate delays
(README.md:181-211, :258-287): grid points are independent, one Float64 comes back per point.
Here the G delay vectors are cut into `world` contiguous blocks (equal cost per point at fixed
N), every rank evaluates its block on its own GPU with NO data-path collective, and ONE
all_gather collects the log-likelihoods (and info codes) for getprobabilities."""
import numpy as np


def shard_bounds(G, world, rank):
    """Static contiguous block partition of range(G): the first G % world ranks get one more."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    base, extra = divmod(G, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def sharded_loglik(evaluate, delays, alpha, rho, group=None, device=None):
    """Evaluates objective on this rank's block of the grid and all-gathers the full vectors.

    evaluate(delays_block, alpha_block, rho_block) -> (loglik_block, info_block); in the product it
    is Objective.loglik_batch.  delays/alpha: (G, L); rho: (G,).  Returns (loglik[G], info[G]) on
    every rank."""
    import torch
    import torch.distributed as dist

    delays = np.ascontiguousarray(np.atleast_2d(delays), dtype=np.float64)
    alpha = np.ascontiguousarray(np.atleast_2d(alpha), dtype=np.float64)
    rho = np.ascontiguousarray(np.atleast_1d(rho), dtype=np.float64)
    G = len(rho)
    if not (dist.is_available() and dist.is_initialized()):
        return evaluate(delays, alpha, rho)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    lo, hi = shard_bounds(G, world, rank)
    ll, info = evaluate(delays[lo:hi], alpha[lo:hi], rho[lo:hi]) if hi > lo else (np.empty(0), np.empty(0, np.int32))
    cap = (G + world - 1) // world  # equal-size payload: [loglik | info] padded
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    mine = torch.zeros(2 * cap, dtype=torch.float64, device=device)
    mine[:hi - lo] = torch.as_tensor(np.asarray(ll, dtype=np.float64), device=device)
    mine[cap:cap + hi - lo] = torch.as_tensor(np.asarray(info, dtype=np.float64), device=device)
    gathered = torch.empty(world * 2 * cap, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(gathered, mine, group=group)   # the single collective of the path
    gathered = gathered.cpu().numpy().reshape(world, 2, cap)
    out_ll, out_info = np.empty(G), np.empty(G, dtype=np.int32)
    for r in range(world):
        a, b = shard_bounds(G, world, r)
        out_ll[a:b] = gathered[r, 0, :b - a]
        out_info[a:b] = gathered[r, 1, :b - a].astype(np.int32)
    return out_ll, out_info


def sharded_grid_fit(fit_block, candidatedelays, group=None, device=None):
    """The per-delay FIT (gpcc_grid_loglik) over the GPUs of a node.  Nelder-Mead iteration counts differ a
    little from delay to delay and neighbouring delays behave alike, so the rows of candidatedelays (G, L) are
    dealt round-robin (rank r takes rows r, r + world, ...), fitted with no data-path
    collective, and ONE all_gather returns [loglik | alpha | rho | info] rows to every rank.

    fit_block(cand_block) -> (loglik[g], alpha[g, L], rho[g], info[g]); in the product
    `lambda c: obj.grid_loglik(c, iterations, ...)[:4]`.  Returns (loglik[G], alpha[G, L], rho[G], info[G])."""
    import torch
    import torch.distributed as dist

    cand = np.ascontiguousarray(np.atleast_2d(candidatedelays), dtype=np.float64)
    G, L = cand.shape
    if not (dist.is_available() and dist.is_initialized()):
        ll, alpha, rho, info = fit_block(cand)
        return np.asarray(ll), np.asarray(alpha), np.asarray(rho), np.asarray(info, dtype=np.int32)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine_rows = cand[rank::world]
    g = len(mine_rows)
    cap, K = (G + world - 1) // world, L + 3
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else "cpu"
    block = np.zeros((cap, K))
    if g:
        ll, alpha, rho, info = fit_block(mine_rows)
        block[:g, 0], block[:g, 1:1 + L], block[:g, 1 + L], block[:g, 2 + L] = ll, alpha, rho, info
    mine = torch.as_tensor(block.ravel(), device=device)
    gathered = torch.empty(world * cap * K, dtype=torch.float64, device=device)
    dist.all_gather_into_tensor(gathered, mine, group=group)   # the single collective of the path
    gathered = gathered.cpu().numpy().reshape(world, cap, K)
    out = np.empty((G, K))
    for r in range(world):
        n_r = len(range(r, G, world))
        out[r::world] = gathered[r, :n_r]
    return out[:, 0].copy(), out[:, 1:1 + L].copy(), out[:, 1 + L].copy(), out[:, 2 + L].astype(np.int32)
