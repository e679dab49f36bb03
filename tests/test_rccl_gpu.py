"""The one collective the north star specifies -- an RCCL all-reduce of the flat gradient buffer between the backward and
the Adam launch -- executed on ROCm.  A 1-GPU box cannot hold two RCCL ranks (RCCL refuses two ranks on one device), so:

* test_rccl_world1_*: a fresh child process opens a ONE-rank "nccl" group and forces the Trainer's collective
  (Trainer(force_collective=True) / ALVQ_FORCE_COLLECTIVE=1); eager and captured steps, one span and two spans.  The
  parameters after 3 steps must be bit-identical to the run with no collective, with exactly 1 (or 2) all_reduce calls
  per step over exactly the flat buffer.
* test_two_ranks_on_one_card_gloo: the N > 1 control flow of bench.py end to end, two ranks sharing the card and
  exchanging gradients over gloo (what tools/rehearse_ranks.sh does by hand), both bucket settings, same final loss.

N > 1 over RCCL stays unmeasured until the driver's 8-GPU run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _env(**extra):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("ALVQ_WIDE_MIN_TILES", None)
    env.update(extra)
    return env


@pytest.mark.parametrize("mode", ["x3mx_hb", "bf16"])
def test_rccl_world1_forced_collective_is_executed_and_exact(mode):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_world1.py"), mode], env=_env(RANK="0", WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RCCL_WORLD1 ")][-1]
    r = json.loads(line[len("RCCL_WORLD1 "):])
    print(json.dumps(r))
    assert r["backend"] == "nccl" and r["world"] == 1 and r["scratch_identity"]
    for name, v in r["variants"].items():
        want = 2 if name.endswith("buckets2") else 1
        assert v["allreduce_calls_per_step"] == want, (name, v)
        assert v["params_bit_identical"] and v["losses_equal"] and v["finite"], (name, v)
        if want == 1:
            assert v["first_call_numels"][0] == v["flat_numel"], (name, v)      # the WHOLE flat buffer in one call
        else:
            assert sum(v["first_call_numels"]) == v["flat_numel"], (name, v)    # two spans that partition it


def test_graph_capture_survives_the_process_group_watchdog():
    """ProcessGroupNCCL's watchdog thread polls the events of earlier collectives (hipEventQuery) while the Trainer captures
    its graphs; under torch's default capture error mode that invalidated about one capture in ten (round 3).  Deterministic
    form: a helper thread calls Event.query() in a tight loop for the whole duration of ONE capture (next to a live one-rank
    nccl group), the capture must succeed, replay, and have been taken in "thread_local" mode."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "rccl_capture_stress.py")],
                       env=_env(RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("RCCL_CAPTURE_SPIN ")][-1][len("RCCL_CAPTURE_SPIN "):])
    print(json.dumps(r))
    assert r["capture_mode"] == "thread_local" and not r["spinner_errors"], r
    assert r["polls_during_capture"] > 100 and r["replayed_loss_finite"], r       # the other thread really was polling meanwhile


def _bench(cmd, env):
    """Run bench.py; returns (compact line = the LAST stdout line, full result from bench_detail.json)."""
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    last = p.stdout.strip().splitlines()[-1]
    assert len(last) < 6000, len(last)                                           # what the driver's 8 KB capture must hold
    with open(os.path.join(ROOT, "bench_detail.json")) as fh:
        return json.loads(last), json.load(fh)


def test_bench_multi_gpu_flow_rehearsed_over_rccl_at_world_1():
    """bench.py's N > 1 code path end to end on one GPU: a one-rank RCCL process group, every mode's Trainer broadcasting its
    parameters, capturing its graphs next to the watchdog thread, issuing the forced all-reduce between the graph replays and
    the Adam launch, the max-over-ranks reduction of the timing, both gradient-exchange settings, the `rccl` block (HIP events
    around the collective alone, the flat-buffer checksum gathered over the ranks) -- and the same final loss as the run
    without a process group."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--batch", "8", "--no-cpu-baseline",
            "--no-parity", "--no-kernel-timer"]
    outs = {}
    for force in ("1", "0"):
        outs[force] = _bench(base, _env(ALVQ_FORCE_COLLECTIVE=force, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    (f, ffull), (b, bfull) = outs["1"], outs["0"]
    assert f["allreduce_calls_per_step"] == 1 and b["allreduce_calls_per_step"] == 0
    assert f["grad_exchange"]["two_spans"]["allreduce_calls_per_step"] == 2 and "grad_exchange" not in b
    assert f["launch"] == "hipGraph replay" and f["final_loss"] == b["final_loss"] and f["dtype"] == "x3mx_hb"
    r = f["rccl"]
    assert r["world"] == 1 and r["nranks_seen"] == 1 and r["backend"].startswith("nccl") and r["ranks_bit_identical"] is True
    assert r["allreduce_bytes"] == 4 * (16836937 + 64 * 18) or r["allreduce_bytes"] > 4 * 16836937      # flat buffer incl. header / padding
    assert r["allreduce_ms"] > 0 and r["per_rank_ms_per_step"]["min"] <= r["per_rank_ms_per_step"]["max"]
    assert "rccl" not in b
    assert set(ffull["modes"]) == {"x3mx_hb", "f16mx_hb", "bf16x3_hb", "f32", "bf16"}
    for m in ffull["modes"]:
        assert ffull["modes"][m]["allreduce_calls_per_step"] == 1, m
        assert ffull["modes"][m]["final_loss"] == bfull["modes"][m]["final_loss"], m


def test_two_ranks_on_one_card_gloo():
    finals = {}
    for buckets in ("1", "2"):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "2",
               "--batch", "8", "--no-secondary", "--no-kernel-timer", "--no-cpu-baseline", "--no-parity"]
        line, _ = _bench(cmd, _env(ALVQ_BENCH_BACKEND="gloo", ALVQ_GRAD_BUCKETS=buckets))
        assert line["n_gpus"] == 2 and line["allreduce_calls_per_step"] == int(buckets) and line["config"]["global_batch"] == 16
        r = line["rccl"]                 # the N > 1 block, here over gloo: both ranks seen, parameters identical on both after the steps
        assert r["world"] == 2 and r["nranks_seen"] == 2 and r["backend"] == "gloo" and r["ranks_bit_identical"] is True
        assert r["allreduce_ms"] > 0 and r["bus_GBps"] > 0 and r["per_rank_ms_per_step"]["min"] <= r["per_rank_ms_per_step"]["max"]
        finals[buckets] = line["final_loss"]
    assert finals["1"] == finals["2"], finals            # same arithmetic whichever way the buffer is reduced


@pytest.mark.parametrize("buckets,launch", [(1, "eager"), (2, "eager"), (1, "graph"), (2, "graph")])
def test_a_step_saturated_on_one_rank_is_skipped_on_every_rank(buckets, launch):
    """The skip verdict is element 0 of the flat gradient buffer and travels through the step's all-reduce: a NaN in ONE rank's
    batch must leave the parameters of BOTH ranks untouched (and identical), count one skipped step on both, and the clean steps
    after it must move both in lockstep -- single all-reduce and two spans (the slot belongs to the span reduced last), eager
    and graph replay.  Two ranks sharing the card over gloo, default mode."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "helpers", "ddp_skip.py"), "x3mx_hb", str(buckets), launch]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("DDP_SKIP ")][-1][len("DDP_SKIP "):])
    print(json.dumps(res))
    assert len(res) == 2
    for r in res:
        assert r["untouched"] and r["slot"] == 1.0 and r["identical_after_skip"], r
        assert r["moved"] and r["identical_at_end"] and r["finite"], r
        assert r["skipped"] == 1 and r["applied"] == (4 if launch == "eager" else 5) - 1, r      # graph: one warm-up step more


@pytest.mark.parametrize("mode,tol,ranks", [("f32", 2e-6, 2), ("x3mx_hb", 2e-5, 2), ("f32", 2e-6, 4)])
@pytest.mark.parametrize("buckets", [1, 2])
def test_two_rank_steps_equal_one_rank_on_the_concatenated_batch(mode, tol, ranks, buckets, tmp_path):
    """Data parallelism through the real HIP path: two (or four) ranks (sharing the card over gloo), each on its share of a batch of 8,
    must end three train steps on the parameters ONE process reaches on the whole batch -- the flat buffer is summed once per
    step, the 1/world factor lives in the Adam launch, local means of equal shards average to the global mean.  Measured:
    f32 losses 7e-8, parameters within 3e-6 (fp32 summation order); f16mx_hb losses 2e-7, parameters within 1e-4 with 0.04 % of
    the entries beyond 1e-5 (each rank picks its own loss scales and meets its own ReLU-edge cases)."""
    import torch
    helper = os.path.join(ROOT, "tests", "helpers", "ddp_equiv.py")
    one, two = str(tmp_path / "one.pt"), str(tmp_path / "two.pt")
    p = subprocess.run([sys.executable, helper, mode, one, str(buckets)], env=_env(), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), helper, mode, two, str(buckets)]
    p = subprocess.run(cmd, env=_env(), capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    a, b = torch.load(one, weights_only=True), torch.load(two, weights_only=True)
    dl = float((a["losses"] - b["losses"]).abs().max() / a["losses"].abs().max())
    moved = (a["flat"] - b["flat"]).abs()
    frac = float((moved > 1e-5).float().mean())
    print("%s ranks %d buckets %d: losses rel %.2e, parameters max |diff| %.2e, fraction beyond 1e-5: %.2e" % (mode, ranks, buckets, dl, float(moved.max()), frac))
    assert dl < tol
    # Adam moves an entry by ~lr per step whatever the gradient's size: an entry whose gradient is rounding noise on one side can
    # differ by a full step; the statement is about how FEW do (f32: none)
    assert float(moved.max()) <= (2e-5 if mode == "f32" else 3.1e-3) and frac < (1e-6 if mode == "f32" else 5e-3)
