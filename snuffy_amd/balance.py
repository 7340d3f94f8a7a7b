"""Length-aware work assignment for the bag-parallel path ("balance by N, longest-processing-time first"; round 6).

A bag's cost is proportional to its patch count N.  With W ranks stepping together -- every training step ends in ONE all-reduce
of the MIL gradients (reference train.py:249-262 stepped per bag; W bags per step is the north-star's data-parallel form) -- a step
costs the LONGEST of its W bags, so on a long-tailed slide set (CAMELYON16: lognormal, sigma ~ 0.5) round-robin over a shuffle pays
E[max of W] / E[mean] ~ 1.8 at W = 8.  Two pure functions, identical on every rank (no communication):

  * step_groups      training: bags sorted by length, cut into groups of W neighbours, the GROUPS visited in shuffled order -- every
                     bag exactly once per epoch, every step made of bags of (nearly) one length;
  * lpt_assignment   evaluation / extraction (no collective in the data path): longest-processing-time-first greedy, the classic
                     4/3-approximation of the minimum makespan.
"""
import numpy as np


def step_groups(lengths, world, order):
    """Visiting order of one training epoch on `world` ranks.

    lengths [n]: patches per bag.  order [n]: this epoch's shuffle (a permutation, the same on every rank -- Trainer._epoch_order).
    Returns pos [steps * world] int64: step s, rank r trains bag pos[s * world + r].  Bags are ranked by (length, position in the
    shuffle), cut into groups of `world` neighbours; the groups are visited in the order of their first member in the shuffle, and
    inside a group the members keep their shuffle order (so rank r does not always get the longest).  n % world != 0: the last,
    short group is filled with the bags that follow it in the length ranking, wrapping (they are visited twice, as in the unbalanced
    path's wrap-around)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    order = np.asarray(order, dtype=np.int64)
    n = len(lengths)
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    rank_in_shuffle = np.empty(n, dtype=np.int64)
    rank_in_shuffle[order] = np.arange(n)
    by_len = np.lexsort((rank_in_shuffle, lengths))                 # primary: length, ties: shuffle position
    steps = (n + world - 1) // world
    padded = np.concatenate([by_len, by_len[: steps * world - n]]) if steps * world > n else by_len
    groups = padded.reshape(steps, world)
    # members in shuffle order; groups in the order of their earliest member
    groups = np.take_along_axis(groups, np.argsort(rank_in_shuffle[groups], axis=1, kind="stable"), axis=1)
    groups = groups[np.argsort(rank_in_shuffle[groups[:, 0]], kind="stable")]
    return groups.reshape(-1)


def lpt_assignment(lengths, world):
    """Longest-processing-time-first: bags by descending length (ties: ascending index), each to the least-loaded rank (ties: lowest
    rank).  Returns a list of `world` index lists, each in ascending bag order.  Deterministic: every rank computes the same."""
    lengths = np.asarray(lengths, dtype=np.int64)
    loads = [0] * world
    mine = [[] for _ in range(world)]
    for i in np.lexsort((np.arange(len(lengths)), -lengths)):
        r = min(range(world), key=lambda q: (loads[q], q))
        loads[r] += int(lengths[i])
        mine[r].append(int(i))
    return [sorted(m) for m in mine]


def imbalance(lengths, groups_or_assignment, world, stepped):
    """max / mean cost ratio.  stepped: pos array of step_groups (a step costs its longest bag): sum_s max_r N / (sum N / world);
    else an lpt_assignment: max_r sum N / (sum N / world)."""
    lengths = np.asarray(lengths, dtype=np.float64)
    if stepped:
        pos = np.asarray(groups_or_assignment, dtype=np.int64).reshape(-1, world)
        return float(lengths[pos].max(axis=1).sum() / (lengths[pos].sum() / world))
    loads = [float(lengths[np.asarray(m, dtype=np.int64)].sum()) if len(m) else 0.0 for m in groups_or_assignment]
    return float(max(loads) / (sum(loads) / world))


def bag_lengths(feats):
    """Patch counts of a list of bags ([N, D] arrays or [1, N, D] tensors)."""
    return np.array([int(f.shape[-2]) for f in feats], dtype=np.int64)
