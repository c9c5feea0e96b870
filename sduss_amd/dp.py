"""Data-parallel placement of requests over replicas (one process + one full replica per GPU, no collective on the data path).

The reference places each waiting request on the dp_rank with the least outstanding sum(resolution^2)
(sduss/dispatcher/policy/greedy.py:16-36; workload = RequestPool.get_pixels_all_dp_rank, dispatcher/request_pool.py:95-102);
for a fixed-resolution stream that degenerates to round-robin, which is what BASELINE.json's north_star names.  The only
cross-rank traffic of the benchmark is bookkeeping: a barrier, the max of the timed region, and a gather of latencies
(``torch.distributed`` over RCCL on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple


def greedy_assign(resolutions: Sequence[int], n_ranks: int, outstanding: Dict[int, int] = None) -> List[int]:
    """dp_rank for every request, in arrival order: argmin of outstanding pixels, then add resolution^2 to it
    (greedy.py:26-34; ties go to the lowest rank as ``min(dict, key=dict.get)`` does on an insertion-ordered dict)."""
    load = {r: 0 for r in range(n_ranks)}
    if outstanding:
        load.update(outstanding)
    out = []
    for res in resolutions:
        target = min(load, key=load.get)
        load[target] += int(res) ** 2
        out.append(target)
    return out


class GreedyPlacer:
    """Stateful mirror of the reference dispatcher's bookkeeping for one node: ``add`` places newly arrived requests
    (GreedyDispath.dispatch_requests, greedy.py:16-36, over RequestPool.get_pixels_all_dp_rank, request_pool.py:95-102),
    ``finish`` drops completed ones (RequestPool.remove_requests).  Placement is sticky for a request's lifetime.
    Checked against the reference classes themselves by tests/test_ref_fixtures.py (tests/golden/ref_greedy_dispatch.json)."""

    def __init__(self, n_ranks: int):
        self.n_ranks = n_ranks
        self._req: Dict[int, Tuple[int, int]] = {}          # request id -> (dp_rank, resolution)

    def outstanding(self) -> Dict[int, int]:
        load = {r: 0 for r in range(self.n_ranks)}
        for rank, res in self._req.values():
            load[rank] += res * res
        return load

    def add(self, request_ids: Sequence[int], resolutions: Sequence[int]) -> List[int]:
        ranks = greedy_assign(resolutions, self.n_ranks, self.outstanding())
        for rid, res, rank in zip(request_ids, resolutions, ranks):
            if rid in self._req:
                raise RuntimeError(f"Request with id {rid} already exists.")    # request_pool.py:40-41
            self._req[rid] = (rank, int(res))
        return ranks

    def finish(self, request_ids: Sequence[int]) -> None:
        for rid in request_ids:
            self._req.pop(rid)


def my_share(n_requests: int, rank: int, world: int, resolutions: Sequence[int] = None) -> List[int]:
    """indices of the requests this rank serves under STATIC pixel balancing: greedy_assign over all requests at once, nothing ever
    finishing (for one resolution: round-robin).  The dynamic placement the reference's dispatcher produces is ``replay_placement``."""
    res = list(resolutions) if resolutions is not None else [1024] * n_requests
    assign = greedy_assign(res, world)
    return [i for i, r in enumerate(assign) if r == rank]


def replay_placement(arrivals: Sequence[float], resolutions: Sequence[int], steps: Sequence[int], world: int,
                     step_seconds: Dict[int, float], max_batch: int) -> List[int]:
    """dp_rank of every request when the arrival trace is REPLAYED through the reference's dispatcher bookkeeping (GreedyPlacer: add on
    arrival, finish on completion), so that the outstanding-pixel loads the placement sees fall as requests complete -- what
    Dispatcher.dispatch / process_worker_outputs do at run time (dispatcher/dispatcher.py:52-118).  Completion times come from a
    deterministic service model (a replica steps at most `max_batch` requests together, FCFS; a step costs the largest per-resolution
    `step_seconds` present, every request in it advances one step), so every rank computes the same table without talking to the others:
    no collective, no rank-0 broadcast.  The model only has to be monotone in load to reproduce the policy's behaviour (least outstanding
    pixels wins); it does not set any reported latency."""
    n = len(arrivals)
    order = sorted(range(n), key=lambda i: arrivals[i])
    placer = GreedyPlacer(world)
    out = [0] * n
    queue = {r: [] for r in range(world)}        # per replica: [request, remaining steps] in arrival order
    clock = {r: 0.0 for r in range(world)}

    def advance(rank: int, until: float) -> None:
        q = queue[rank]
        while q:
            batch = q[:max_batch]
            dt = max(step_seconds.get(int(resolutions[i]), max(step_seconds.values())) for i, _ in batch)
            start = max(clock[rank], arrivals[batch[0][0]])
            if start + dt > until:
                break
            clock[rank] = start + dt
            for e in batch:
                e[1] -= 1
            done = [e[0] for e in q if e[1] <= 0]
            if done:
                placer.finish(done)
                q[:] = [e for e in q if e[1] > 0]
    for i in order:
        for r in range(world):
            advance(r, arrivals[i])
        rank = placer.add([i], [int(resolutions[i])])[0]
        out[i] = rank
        queue[rank].append([i, int(steps[i])])
    return out


class FcfsMixed:
    """Mirror of the per-cycle batching decisions of the reference's worker scheduler under its FCFS_Mixed policy
    (sduss/worker/scheduler/policy/FCFS_Mixed.py:25-76 inside scheduler.py:56-166 over request_pool.py): every cycle the OLDEST unfinished
    request names the stage (PREPARE -> DENOISING -> POSTPROCESSING), and the cycle runs the `max_num` oldest requests that are in that same
    stage, grouped by resolution in age order; is_sliced / patch_size are forced to True / 256 (FCFS_Mixed.py:69-70).  Consequences the
    mirror reproduces: a request that arrives while older ones denoise waits in PREPARE until it is the oldest (batches drain before new
    ones form), and a request that finished its steps waits in POSTPROCESSING until it is the oldest.  ``update`` is
    Scheduler.update_reqs_status + process_output: PREPARE -> DENOISING; one step less, POSTPROCESSING at zero; postprocessed requests
    leave.  Checked cycle by cycle against the reference classes themselves: tests/golden/ref_fcfs_mixed.json, tests/test_ref_fixtures.py."""
    PREPARE, DENOISING, POSTPROCESSING = "PREPARE", "DENOISING", "POSTPROCESSING"

    def __init__(self, max_num: int):
        self.max_num = max_num
        self._req: Dict[int, list] = {}                      # id -> [arrival, resolution, status, remaining steps]; insertion order = pool order

    def add(self, rid: int, arrival: float, resolution: int, steps: int) -> None:
        if rid in self._req:
            raise RuntimeError(f"WorkerRequest with id {rid} already exists.")      # request_pool.py:39-40
        self._req[rid] = [float(arrival), int(resolution), self.PREPARE, int(steps)]

    def has_unfinished(self) -> bool:
        return bool(self._req)

    def schedule(self):
        """(status, {resolution: [ids]}, is_sliced, patch_size) of the next cycle"""
        order = sorted(self._req, key=lambda i: self._req[i][0])     # stable: ties keep pool order, as the reference's sort does
        status = self._req[order[0]][2]
        chosen: Dict[int, List[int]] = {}
        for rid in [i for i in order if self._req[i][2] == status][:self.max_num]:
            chosen.setdefault(self._req[rid][1], []).append(rid)
        return status, chosen, True, 256

    def update(self, decision) -> List[int]:
        """apply a cycle's effect; returns the ids that finished (left the pool)"""
        status, chosen = decision[0], decision[1]
        ids = [i for v in chosen.values() for i in v]
        if status == self.PREPARE:
            for i in ids:
                self._req[i][2] = self.DENOISING
        elif status == self.DENOISING:
            for i in ids:
                self._req[i][3] -= 1
                if self._req[i][3] == 0:
                    self._req[i][2] = self.POSTPROCESSING
        else:
            for i in ids:
                del self._req[i]
            return ids
        return []


def max_over_ranks(value: float, dist=None, device=None) -> float:
    """max of a host scalar over ranks (the timed region of bench.py)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_stream_stats(latencies: List[float], window: Tuple[float, float], dist=None):
    """(all latencies, (first arrival, last finish)) over ranks."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(latencies), window
    gathered = [None] * dist.get_world_size()
    dist.all_gather_object(gathered, (list(latencies), tuple(window)))
    lat = [x for g in gathered for x in g[0]]
    return lat, (min(g[1][0] for g in gathered), max(g[1][1] for g in gathered))
