"""GPU, 2 ranks over RCCL (skipped when fewer than 2 GPUs are visible -- the driver's single-GPU boxes skip it, the 8-GPU scaling node
runs it): FlatTrainer's bucketed gradient all-reduce on the real model.  After each step both ranks hold identical parameters, and the
reduced gradient equals the mean of the two ranks' single-GPU gradients (reference semantics: DDP mean all-reduce, run_rpn.py:235-236)."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(rank):
    g = torch.Generator().manual_seed(500 + rank)
    x = torch.rand(4, 48, 40, 32, generator=g)
    gt = torch.tensor([[14., 12., 12., 10., 8., 9., 0.3], [26., 20., 18., 12., 12., 8., -0.6], [30., 10., 10., 8., 9., 7., 0.1]]) + rank
    return x, gt


def _grads(model, x, gt, dev):
    torch.manual_seed(11)
    _, losses, _ = model([x.to(dev)], [gt.to(dev)])
    (losses["loss_objectness"] + 5.0 * losses["loss_rpn_box_reg"]).backward()


def _worker(rank, world, port, q):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from nerf_rpn_amd.engine import FlatTrainer
    from test_gpu_e2e import build
    model = build(True, 160, dev).train()
    tr = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, bucket_bytes=32 << 20)
    x, gt = _scene(rank)
    _grads(model, x, gt, dev)             # step 0 learns the notification counts (all buckets reduced at the end of backward)
    tr.sync_gradients()
    out = [(tr.flat_grads() / world).cpu()]
    tr.g_arena.zero_()
    _grads(model, x, gt, dev)             # step 1 launches each bucket's all-reduce as soon as its last gradient has landed
    early = sum(tr.launched)
    tr.step()
    q.put((rank, out, tr.flat_params().cpu(), len(tr.buckets), early))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_rccl_gradient_exchange(dev):
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs (RCCL)")
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, ga, pa, nb, early), (_, gb, pb, _, _) = res
    assert nb >= 2 and early >= 1                                       # buckets went out during backward on the second step
    assert torch.equal(pa, pb)                                          # identical parameters after the step on both ranks
    assert torch.equal(ga[0], gb[0])                                    # the all-reduce result is the same tensor on both ranks
    # reference: each rank's gradient computed alone (deterministic kernels), averaged on the host
    from test_gpu_e2e import build
    alone = []
    for rank in range(2):
        m = build(True, 160, dev).train()
        x, gt = _scene(rank)
        _grads(m, x, gt, dev)
        alone.append(torch.cat([p.grad.reshape(-1) for p in m.parameters()]).cpu())
    mean = (alone[0] + alone[1]) / 2
    assert torch.allclose(ga[0], mean, rtol=1e-6, atol=1e-9 + 1e-6 * mean.abs().max().item())


def _single_worker(port, q):
    """One rank over RCCL on the one GPU of the box, the exchange machinery forced on (NRPN_FORCE_EXCHANGE=1): every exchange mode runs
    its real collectives (all_reduce / reduce_scatter + all_gather / all_to_all + all_gather) on the launch stream behind the producer
    events.  With one rank the reduced gradient IS the local gradient: exact for the fp32 modes, bf16-rounded for a2a_bf16."""
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    from nerf_rpn_amd.engine import FlatTrainer
    from test_gpu_e2e import build
    x, gt = _scene(0)

    def run(exchange):
        torch.manual_seed(3)
        model = build(True, 160, dev).train()
        tr = FlatTrainer(model, lr=1e-4, weight_decay=0.01, clip_grad_norm=0.1, bucket_bytes=8 << 20, total_steps=10, exchange=exchange)
        _grads(model, x, gt, dev)
        tr.sync_gradients()
        g0 = tr.flat_grads().cpu()
        tr.step()
        early = 0
        for _ in range(2):
            _grads(model, x, gt, dev)
            early += sum(tr.launched)
            tr.step()
        torch.cuda.synchronize()
        return g0, tr.flat_params().cpu(), len(tr.buckets), early

    out = {"plain": run(None)}                      # no process group yet: the single-process trainer
    os.environ["NRPN_FORCE_EXCHANGE"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    for mode in ("allreduce", "rs_ag", "a2a_bf16"):
        out[mode] = run(mode)
    q.put({k: (v[0].numpy(), v[1].numpy(), v[2], v[3]) for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


def test_single_rank_rccl_exchange_modes(dev):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_worker, args=(_free_port(), q))
    p.start()
    res = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0
    g_plain, p_plain, nb_plain, _ = res["plain"]
    assert nb_plain == 0
    for mode in ("allreduce", "rs_ag"):
        g, prm, nb, early = res[mode]
        assert nb >= 2 and early >= 1, (mode, nb, early)          # buckets exist and went out during the backward of steps 1-2
        assert (g == g_plain).all() and (prm == p_plain).all(), mode    # one rank: the exchange is the identity, bit for bit
    g, prm, nb, early = res["a2a_bf16"]
    assert nb >= 2 and early >= 1
    scale = float(abs(g_plain).max())
    assert abs(g - g_plain).max() <= 2.0 ** -8 * scale and torch.isfinite(torch.from_numpy(prm)).all()


def test_a2a_reduce_kernel_matches_the_torch_formulation(dev):
    """nrpn_a2a_reduce_bf16 (the reduction step of the bf16 all-to-all exchange): fp32 sum over the ranks in ascending order, the own chunk
    from the fp32 bucket, one rounding -- bit-equal to the five-op torch formulation it replaces."""
    from nerf_rpn_amd import ops
    g = torch.Generator().manual_seed(3)
    for world, chunk in ((2, 4096), (8, 64 * 1000 + 8), (4, 16)):
        bucket = torch.randn(world * chunk, generator=g).to(dev)
        recv = torch.randn(world * chunk, generator=g).to(torch.bfloat16).to(dev)
        for rank in (0, world - 1):
            parts = recv.view(world, chunk).float()
            parts[rank] = bucket.view(world, chunk)[rank]
            want = torch.zeros(chunk, device=dev)
            for r in range(world):
                want = want + parts[r]
            got = ops.a2a_reduce(recv, bucket, rank, world)
            assert torch.equal(got, want.to(torch.bfloat16)), (world, chunk, rank)
