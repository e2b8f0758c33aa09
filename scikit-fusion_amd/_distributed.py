"""Sharding of the independent random restarts (``n_run``) over the GPUs of a node.

The reference ships every restart to a joblib worker process (``dfmf.py:87-95``,
``dfmc.py:96-104``, ``dfmf.py:191-199``).  Here the unit of parallelism is one process per GPU
(launched with ``python -m torch.distributed.run``): restart ``k`` runs on rank ``k % world``,
every rank keeps its relations resident on its own GPU, and there is **no collective on the data
path** -- only the fitted ``(G, S)`` of each run are exchanged once at the end
(``all_gather_object``; backend ``nccl`` = RCCL on ROCm, ``gloo`` on CPU for tests).
Single-process use (no process group) degenerates to "all runs here".
"""


def _dist():
    try:
        import torch.distributed as dist
    except ImportError:
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def world():
    """(rank, world_size) of the default process group, (0, 1) without one."""
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d else (0, 1)


def my_runs(n_run):
    """Indices of the restarts this rank computes (round robin)."""
    rank, size = world()
    return [k for k in range(n_run) if k % size == rank]


def gather_runs(local, n_run):
    """``local``: {run index: result} computed by this rank -> list of all results in run order,
    identical on every rank."""
    d = _dist()
    if d is None:
        return [local[k] for k in range(n_run)]
    parts = [None] * d.get_world_size()
    d.all_gather_object(parts, local)
    merged = {}
    for part in parts:
        merged.update(part)
    missing = [k for k in range(n_run) if k not in merged]
    if missing:
        raise RuntimeError('restarts %r were not computed by any rank' % missing)
    return [merged[k] for k in range(n_run)]


def partition_relations(rel_list, theta_list, n_obj, rank_of):
    """Deterministic assignment of relations and constraints to ranks (longest processing time
    first on the contraction cost n_i*n_j*(c_i+c_j), constraints n_i^2*c_i).  Every rank computes
    the same table.  Returns (relation owner per index, constraint owner per index)."""
    _, size = world()
    load = [0.0] * size
    jobs = []
    for k, (i, j, _, _) in enumerate(rel_list):
        jobs.append((float(n_obj[i]) * n_obj[j] * (rank_of[i] + rank_of[j]), 0, k))
    for k, (i, _) in enumerate(theta_list):
        jobs.append((2.0 * float(n_obj[i]) ** 2 * rank_of[i], 1, k))
    rel_owner, theta_owner = [0] * len(rel_list), [0] * len(theta_list)
    for cost, kind, k in sorted(jobs, key=lambda t: (-t[0], t[1], t[2])):
        r = min(range(size), key=lambda q: (load[q], q))
        load[r] += cost
        (theta_owner if kind else rel_owner)[k] = r
    return rel_owner, theta_owner


def partition_rows(rel_list, theta_list, n_obj, rank_of, align=256, size=None):
    """Balanced row-block partition ("relations and, for balance, row blocks of large
    relations"): the relations are laid end to end, every row weighted by its contraction cost
    n_j*(c_i+c_j), and cut into `size` stretches of equal cost at row boundaries that are multiples
    of `align`.  A rank therefore holds at most one contiguous block of any relation.  Returns
    (blocks, theta_owner): blocks[k] = [(rank, row_begin, n_rows), ...] covering relation k in row
    order (the first block's rank also adds the column-side terms); every rank computes the same
    table."""
    if size is None:
        _, size = world()
    row_cost = [float(n_obj[j]) * (rank_of[i] + rank_of[j]) for i, j, _, _ in rel_list]
    total = sum(c * n_obj[rel[0]] for c, rel in zip(row_cost, rel_list))
    share = total / float(size) if size else 0.0
    blocks = [[] for _ in rel_list]
    load = [0.0] * size
    q = 0
    for k, (i, j, _, _) in enumerate(rel_list):
        n_i, done = int(n_obj[i]), 0
        while done < n_i:
            left = n_i - done
            room = share - load[q]
            take = left
            if q < size - 1 and room < left * row_cost[k]:
                take = int(room / row_cost[k]) // align * align
                if left - take < align:          # do not leave a sliver for the next rank
                    take = left
            if take <= 0:                        # this rank is full
                q += 1
                continue
            blocks[k].append((q, done, take))
            load[q] += take * row_cost[k]
            done += take
            if q < size - 1 and load[q] >= share - 0.5 * align * row_cost[k]:
                q += 1
    theta_owner = [0] * len(theta_list)
    jobs = sorted(((2.0 * float(n_obj[i]) ** 2 * rank_of[i], k) for k, (i, _) in enumerate(theta_list)),
                  key=lambda t: (-t[0], t[1]))
    for cost, k in jobs:
        r = min(range(size), key=lambda x: (load[x], x))
        load[r] += cost
        theta_owner[k] = r
    return blocks, theta_owner


def gather_backbones(local, n_rel):
    """{relation index: S} of this rank -> list of all backbones, identical on every rank."""
    return gather_runs(local, n_rel)


def sum_over_ranks(values):
    """Element-wise sum of a list of floats over the ranks (identical result on every rank): the per-relation
    squared errors of a sharded fit, each rank contributing what it holds (reference _dfmf.py:301-319 needs the
    error of every relation for `compute_err` / `stopping*`)."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return [float(v) for v in values]
    import torch
    on_gpu = d.get_backend() != 'gloo' and torch.cuda.is_available()
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device='cuda' if on_gpu else 'cpu')
    d.all_reduce(t, op=d.ReduceOp.SUM)
    return [float(v) for v in t.cpu().tolist()]


def same_on_all_ranks(value):
    """Rank 0's `value` on every rank (a picklable host decision that fixes a layout or a partial-sum convention for the
    whole group -- e.g. whether a masked relation is kept as lists of its known entries: each rank would otherwise read
    its own environment); the value itself without a group."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return value
    box = [value]
    d.broadcast_object_list(box, src=0)
    return box[0]
