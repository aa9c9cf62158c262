"""CPU, world sizes 2 / 3 / 4 / 8 (gloo): the data-parallel gradient exchange of nerf_rpn_amd.engine.FlatTrainer -- rank-0 weight
broadcast, bucketed SUM all-reduce launched from post-accumulate hooks, unused-parameter buckets, 1/world folding.
(The optimiser kernels themselves are GPU-only and are covered by tests/test_gpu_conv.py::test_layout_roundtrip_and_adamw.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_ranks(target, extra_args, world=2, attempts=3):
    """Spawn ``world`` ranks of ``target(rank, world, port, *extra_args, queue)`` and return their results sorted by rank.  The rendezvous
    port is picked by binding port 0 and released before the ranks bind it, so another process can grab it in between (seen as a rare
    gloo connection error): a failed RENDEZVOUS is retried on a fresh port; results themselves are never retried into agreement --
    every assertion is made by the caller on the one set of results returned."""
    import queue as _queue
    ctx = mp.get_context("spawn")
    last = None
    for _ in range(attempts):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=target, args=(r, world, port, *extra_args, q)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
        except _queue.Empty as e:
            res, last = None, e
        for p in procs:
            p.join(timeout=60)
            if p.is_alive():
                p.kill()
        if res is not None and all(p.exitcode == 0 for p in procs):
            return res
        last = last or RuntimeError(f"rank exit codes {[p.exitcode for p in procs]}")
    raise AssertionError(f"the {world}-rank run failed {attempts} times: {last}")


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 16)
        self.unused = nn.Linear(4, 4)      # never touched by forward: its bucket must still be reduced (zeros)
        self.c = nn.Linear(16, 3)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _worker(rank, world, port, bucket_bytes, exchange, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if os.path.exists("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # the container hostname may not resolve: keep gloo's transport on loopback
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if isinstance(exchange, tuple):          # several modes in one spawn (world sizes 4 / 8: the spawn dominates the test's time)
        out = [_exchange_rounds(rank, world, bucket_bytes, mode) for mode in exchange]
        q.put((rank, out))
    else:
        q.put((rank, *_exchange_rounds(rank, world, bucket_bytes, exchange)))      # by value: a shared-memory tensor handle dies with this process
    dist.barrier()
    dist.destroy_process_group()


def _exchange_rounds(rank, world, bucket_bytes, exchange):
    from nerf_rpn_amd.engine import FlatTrainer
    torch.manual_seed(100 + rank)            # different init per rank: the trainer must broadcast rank 0's weights
    model = Tiny()
    tr = FlatTrainer(model, bucket_bytes=bucket_bytes, exchange=exchange)
    w0 = tr.p_arena.clone()
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(8, 8, generator=g), torch.randn(8, 3, generator=g)
    per = 8 // world
    xs, ys = x_all[rank * per:(rank + 1) * per], y_all[rank * per:(rank + 1) * per]
    early = []
    for _ in range(3):                       # three rounds: bucket counters must re-arm; from the second on buckets go out during backward
        tr.g_arena.zero_()
        ((model(xs) - ys) ** 2).sum().backward()
        early.append(sum(tr.launched))
        tr.sync_gradients()
    return w0.numpy(), (tr.flat_grads() / world).numpy(), len(tr.buckets), early


@pytest.mark.parametrize("exchange", ["allreduce", "rs_ag", "a2a_bf16"])
@pytest.mark.parametrize("bucket_bytes", [64, 1 << 20])
def test_flat_trainer_gradient_exchange(bucket_bytes, exchange):
    """every exchange mode leaves the mean of the per-rank gradients in every rank's arena: fp32 all-reduce and reduce-scatter +
    all-gather exactly, the bf16 all-to-all form within bf16 rounding of the REMOTE contributions (fp32 accumulation, own chunk in fp32)."""
    res = _run_ranks(_worker, (bucket_bytes, exchange))
    (_, w_a, g_a, nb, _), (_, w_b, g_b, _, _) = res
    w_a, g_a, w_b, g_b = (torch.from_numpy(t) for t in (w_a, g_a, w_b, g_b))
    assert torch.equal(w_a, w_b)                             # rank 0's weights everywhere
    assert torch.equal(g_a, g_b) if exchange != "allreduce" else torch.allclose(g_a, g_b)      # identical reduced gradients on every rank
    assert nb >= (4 if bucket_bytes == 64 else 1)
    # reference: single process, whole batch, same (rank-0) weights; mean over ranks of per-rank sums = sum/2
    torch.manual_seed(100)
    ref = Tiny()
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(8, 8, generator=g), torch.randn(8, 3, generator=g)
    ((ref(x_all) - y_all) ** 2).sum().backward()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref.parameters()]) / 2
    if exchange == "a2a_bf16":     # remote chunks and the reduced chunk travel as bf16: 2^-8 of the per-rank contributions' scale
        assert torch.allclose(g_a, flat, rtol=2 ** -7, atol=2 ** -8 * flat.abs().max().item())
    else:
        assert torch.allclose(g_a, flat, atol=1e-5)


def _reference_mean_gradient(world):
    torch.manual_seed(100)
    ref = Tiny()
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(8, 8, generator=g), torch.randn(8, 3, generator=g)
    ((ref(x_all) - y_all) ** 2).sum().backward()
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref.parameters()]) / world


@pytest.mark.parametrize("world", [4, 8])
def test_gradient_exchange_at_world_sizes_4_and_8(world):
    """every exchange mode at the node shapes the scaling run uses (4 and 8 ranks; gloo on the CPU stands in for RCCL): rank 0's
    weights everywhere, the mean of the per-rank gradients in every rank's arena, many small buckets (64 B: one per parameter) so that the
    chunked modes split buckets of 64 floats into 4 / 8 chunks, and buckets launched during backward from the second step on."""
    modes = ("allreduce", "rs_ag", "a2a_bf16")
    res = _run_ranks(_worker, (64, modes), world=world)
    assert [r[0] for r in res] == list(range(world))
    flat = _reference_mean_gradient(world)
    for k, exchange in enumerate(modes):
        w0 = torch.from_numpy(res[0][1][k][0])
        g0 = torch.from_numpy(res[0][1][k][1])
        for _, per_mode in res:
            w, g, nb, early = per_mode[k]
            assert torch.equal(torch.from_numpy(w), w0), exchange
            g = torch.from_numpy(g)
            assert torch.equal(g, g0) if exchange != "allreduce" else torch.allclose(g, g0), exchange
            assert nb >= 4
            assert early[0] == 0 and early[1] > 0 and early[2] > 0, (exchange, early)
        if exchange == "a2a_bf16":
            assert torch.allclose(g0, flat, rtol=2 ** -7, atol=2 ** -8 * flat.abs().max().item() * 2), exchange
        else:
            assert torch.allclose(g0, flat, atol=1e-5), exchange


def _world3_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if os.path.exists("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    os.environ["NRPN_QUIET"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_rpn_amd.engine import FlatTrainer
    raised = {}
    for mode in ("rs_ag", "a2a_bf16"):
        torch.manual_seed(100 + rank)
        try:
            FlatTrainer(Tiny(), exchange=mode)
            raised[mode] = None
        except ValueError as e:
            raised[mode] = str(e)
    out = {}
    for mode in ("allreduce", "auto"):
        torch.manual_seed(100 + rank)
        model = Tiny()
        tr = FlatTrainer(model, bucket_bytes=64, exchange=mode)
        g = torch.Generator().manual_seed(7)
        x_all, y_all = torch.randn(9, 8, generator=g), torch.randn(9, 3, generator=g)
        xs, ys = x_all[rank * 3:(rank + 1) * 3], y_all[rank * 3:(rank + 1) * 3]
        ((model(xs) - ys) ** 2).sum().backward()
        tr.sync_gradients()
        out[mode] = (tr.exchange, (tr.flat_grads() / world).numpy(), sorted(tr.exchange_table or {}))
    q.put((rank, raised, out))
    dist.barrier()
    dist.destroy_process_group()


def test_chunked_exchange_rejects_world_sizes_that_do_not_divide_a_bucket():
    """rs_ag / a2a_bf16 hand every rank 1/world of a 64-float-aligned bucket.  BEHAVIOUR at a 3-rank group: constructing
    them raises ValueError with a clear message on every rank; 'allreduce' and 'auto' (which then only considers the all-reduce) construct,
    and their exchange leaves the mean gradient everywhere."""
    res = _run_ranks(_world3_worker, (), world=3)
    torch.manual_seed(100)
    ref = Tiny()
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(9, 8, generator=g), torch.randn(9, 3, generator=g)
    ((ref(x_all) - y_all) ** 2).sum().backward()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref.parameters()]) / 3
    for _, raised, out in res:
        for mode in ("rs_ag", "a2a_bf16"):
            assert raised[mode] is not None and "divides 64" in raised[mode] and "world" in raised[mode], raised
        for mode in ("allreduce", "auto"):
            chosen, grads, table = out[mode]
            assert chosen == "allreduce"
            assert torch.allclose(torch.from_numpy(grads), flat, atol=1e-5)
        assert out["auto"][2] and all(k[0] == "allreduce" for k in out["auto"][2])       # the chunked modes were not even timed


def _auto_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if os.path.exists("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_rpn_amd.engine import FlatTrainer, AUTO_MODES
    torch.manual_seed(100 + rank)
    model = Tiny()
    tr = FlatTrainer(model, exchange="auto")           # comm-only measurement at start-up, fastest fp32 (mode, bucket size) wins
    table = dict(tr.exchange_table)
    assert tr.exchange in AUTO_MODES and (tr.exchange, tr.bucket_bytes >> 20) in table        # only the candidates are timed
    assert table[(tr.exchange, tr.bucket_bytes >> 20)] == min(table.values())
    # a later measurement (bench.py's full table, every mode) must leave early bucket launches armed: ADVICE r4 #3
    assert float(tr.g_arena.abs().max()) == 0.0       # the measurement leaves a clean arena
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(8, 8, generator=g), torch.randn(8, 3, generator=g)
    xs, ys = x_all[rank * 4:(rank + 1) * 4], y_all[rank * 4:(rank + 1) * 4]
    ((model(xs) - ys) ** 2).sum().backward()
    tr.sync_gradients()
    full = tr.measure_exchange()                        # after the learning step, as bench.py does
    assert {k[0] for k in full} == {"allreduce", "rs_ag", "a2a_bf16"} and len(full) == 9
    tr._build_buckets(64)                               # one bucket per parameter (the default size puts the unused layer into the only bucket)
    assert any(tr.early), "re-bucketing after the learning step must keep early launches (recomputed from the learned counts)"
    tr.g_arena.zero_()
    ((model(xs) - ys) ** 2).sum().backward()
    launched_early = sum(tr.launched)
    tr.sync_gradients()
    assert launched_early > 0
    q.put((rank, tr.exchange, tr.bucket_bytes, sorted((k[0], k[1], v) for k, v in table.items()), (tr.flat_grads() / world).numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_exchange_auto_measures_every_mode_and_all_ranks_agree():
    """exchange='auto' (the default): every rank times every candidate (fp32 mode, bucket size) on the real arena, takes the MAX over ranks and
    therefore picks the same winner; the exchange that follows still leaves the mean gradient everywhere."""
    res = _run_ranks(_auto_worker, ())
    (_, mode_a, bb_a, tab_a, g_a), (_, mode_b, bb_b, tab_b, g_b) = res
    assert (mode_a, bb_a) == (mode_b, bb_b) and tab_a == tab_b
    assert {m for m, _, _ in tab_a} == {"allreduce", "rs_ag"} and len(tab_a) == 6
    g_a, g_b = torch.from_numpy(g_a), torch.from_numpy(g_b)
    torch.manual_seed(100)
    ref = Tiny()
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(8, 8, generator=g), torch.randn(8, 3, generator=g)
    ((ref(x_all) - y_all) ** 2).sum().backward()
    flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ref.parameters()]) / 2
    assert mode_a in ("allreduce", "rs_ag")
    assert torch.allclose(g_a, flat, atol=1e-5) and torch.allclose(g_b, flat, atol=1e-5)


class _SinkLinear(torch.autograd.Function):
    """CPU stand-in for the HIP backward kernels: accumulates the weight gradient straight into the trainer's arena slot
    (ops.GradSink) and notifies, instead of returning a gradient tensor to autograd."""

    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x, w)
        ctx.sink = getattr(w, "_nrpn_sink", None)
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        gw = dy.t() @ x
        if ctx.sink is not None:
            ctx.sink.slot.add_(gw)
            ctx.sink.notify()
            gw = None
        return dy @ w, gw


class Shared(nn.Module):
    def __init__(self):
        super().__init__()
        self.first = nn.Linear(6, 6)
        self.w = nn.Parameter(torch.randn(6, 6) * 0.3)       # used on THREE "levels" per step, like the RPN head convs
        self.last = nn.Linear(6, 2)

    def forward(self, x):
        h = torch.relu(self.first(x))
        outs = [_SinkLinear.apply(h * s, self.w) for s in (1.0, 0.5, 2.0)]
        return self.last(sum(outs))


def _sink_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if os.path.exists("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_rpn_amd.engine import FlatTrainer
    torch.manual_seed(5)
    model = Shared()
    tr = FlatTrainer(model, bucket_bytes=64)
    g = torch.Generator().manual_seed(11)
    x_all, y_all = torch.randn(6, 6, generator=g), torch.randn(6, 2, generator=g)
    xs, ys = x_all[rank * 3:(rank + 1) * 3], y_all[rank * 3:(rank + 1) * 3]
    early = []
    for it in range(3):      # step 0 learns the notification counts (3 for the shared weight); later steps launch buckets early
        tr.g_arena.zero_()
        ((model(xs) - ys) ** 2).sum().backward()
        early.append(sum(tr.launched))
        tr.sync_gradients()
    q.put((rank, (tr.flat_grads() / world).numpy(), list(tr.expected), early))
    dist.barrier()
    dist.destroy_process_group()


def test_direct_sink_accumulation_with_shared_weights():
    res = _run_ranks(_sink_worker, ())
    (_, g_a, expected, early), (_, g_b, _, _) = res
    g_a, g_b = torch.from_numpy(g_a), torch.from_numpy(g_b)
    assert torch.allclose(g_a, g_b)
    names = [n for n, _ in Shared().named_parameters()]
    # three sink notifications (+ one from autograd's accumulate hook on builds that fire it for an undefined gradient): the count
    # is LEARNED in the first step, which is what makes the early launches safe either way
    assert expected[names.index("w")] in (3, 4) and all(e == 1 for i, e in enumerate(expected) if i != names.index("w"))
    assert early[0] == 0 and early[1] > 0 and early[2] > 0        # buckets go out during backward once the counts are known
    torch.manual_seed(5)
    ref = Shared()
    g = torch.Generator().manual_seed(11)
    x_all, y_all = torch.randn(6, 6, generator=g), torch.randn(6, 2, generator=g)
    ((ref(x_all) - y_all) ** 2).sum().backward()
    flat = torch.cat([p.grad.reshape(-1) for p in ref.parameters()]) / 2
    assert torch.allclose(g_a, flat, atol=1e-5)


def test_one_cycle_matches_torch():
    from nerf_rpn_amd.engine import one_cycle
    p = nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=3e-4)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-4, total_steps=50)
    for step in range(50):
        lr, b1 = one_cycle(step, 50, 3e-4)
        assert abs(lr - opt.param_groups[0]["lr"]) < 1e-12 and abs(b1 - opt.param_groups[0]["betas"][0]) < 1e-12, step
        opt.step()
        if step < 49:
            sched.step()


def _fcos_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if os.path.exists("/sys/class/net/lo"):
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nerf_rpn_amd.model.fcos.loss import FCOSLossComputation
    ev = FCOSLossComputation.__new__(FCOSLossComputation)         # the reductions only: the rest of the loss runs on HIP kernels
    ev.world_size = world
    out = []
    # step 1: rank 0 has 3 positives (centerness sum 1.5), rank 1 none; step 2: 4 and 6 positives; step 3: nobody has any
    for num_pos, ctr in (((3, 1.5), (0, None)), ((4, 2.0), (6, 1.0)), ((0, None), (0, None))):
        n, c = (num_pos, ctr)[rank]
        out.append(ev.normalisers(torch.tensor([n], dtype=torch.int32), None if c is None else torch.tensor(c)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_fcos_loss_normalisers_across_ranks():
    """num_pos / centerness-sum exchange of the data-parallel FCOS loss (reference fcos/loss.py:533-550, 588): both ranks get the same two
    scalars, a rank without positives still joins the second reduction (no hang), and all-empty steps clamp to 1 / 0."""
    res = _run_ranks(_fcos_worker, ())
    (_, a), (_, b) = res
    assert a == b
    assert a[0] == (1.5, 0.75) and a[1] == (5.0, 1.5) and a[2] == (1.0, 0.0)
