"""bench.py's command-line contract that can be checked without a GPU: `--gpus N` launches N ranks itself, the
parent touches no GPU, a rank refuses a mismatching WORLD_SIZE, and a failing rank makes the launcher exit non-zero."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE")}
    e.update(kw)
    return e


def test_gpus_n_spawns_n_ranks_and_propagates_failure():
    # no GPU here: every rank must exit non-zero ("needs device"), and so must the launcher; no JSON line on stdout
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "1"], capture_output=True,
                       text=True, timeout=600, env=_env())
    assert p.returncode != 0
    assert "launching 2 ranks" in p.stderr and "torch.distributed.run" in p.stderr
    assert "needs device" in p.stderr               # a rank ran and refused (torchrun may stop its sibling first)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_rank_refuses_wrong_world_size():
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "5"], capture_output=True, text=True, timeout=300,
                       env=_env(RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert p.returncode == 2 and "refusing" in p.stderr


def test_parent_mode_never_imports_torch_cuda_state():
    # the launcher branch returns before `import torch`: check by running it with a sabotaged torch on the path
    src = open(BENCH).read().split("def main():", 1)[1]
    head = src.split("import torch", 1)[0]
    assert "launch_ranks(args" in head, "the N-rank launcher must run before torch is imported"


def test_dry_launch_shows_the_child_job_and_starts_nothing():
    """`--gpus N --dry-launch`: one JSON line with the exact N-rank command (torch.distributed.run, one process per GPU,
    127.0.0.1 rendezvous, the caller's own flags passed through) and the communication environment the ranks would get
    (HSA_ENABLE_IPC_MODE_LEGACY=0: the host driver only supports dmabuf IPC); exit 0, nothing launched."""
    p = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "20", "--warmup", "5", "--dry-launch"], capture_output=True,
                       text=True, timeout=120, env=_env(NCCL_DEBUG="INFO"))
    assert p.returncode == 0, p.stderr
    assert "launching" not in p.stderr
    j = json.loads(p.stdout.strip())
    cmd = j["cmd"]
    assert j["dry_launch"] and j["n_ranks"] == 8
    # the EXACT child command: the driver's own line for N > 1, nothing more
    port = cmd[cmd.index("--master-port") + 1]
    assert port.isdigit() and 1024 <= int(port) <= 65535
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                   "--master-port", port, BENCH, "--gpus", "8", "--steps", "20", "--warmup", "5"]
    # ... and the communication environment of the ranks: what the parent adds (dmabuf IPC, the loopback rendezvous) plus whatever
    # HSA_ / HIP_ / ROCR_ / NCCL_ / RCCL_ / GPU_ / MASTER_ / CUDA_VISIBLE variables the caller had set — no rank variables (the
    # launcher of torch.distributed.run sets RANK / LOCAL_RANK / WORLD_SIZE per rank)
    want = {k: v for k, v in _env(NCCL_DEBUG="INFO").items()
            if k.startswith(("HSA_", "HIP_", "ROCR_", "NCCL_", "RCCL_", "GPU_", "MASTER_", "TORCH_NCCL", "CUDA_VISIBLE"))}
    want.update(HSA_ENABLE_IPC_MODE_LEGACY=want.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
    assert j["env"] == want, (j["env"], want)
    assert not {"RANK", "LOCAL_RANK", "WORLD_SIZE"} & set(j["env"])
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    tail = cmd[cmd.index(BENCH) + 1:]
    assert tail == ["--gpus", "8", "--steps", "20", "--warmup", "5"]          # --dry-launch itself is not passed on
    assert j["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and j["env"]["MASTER_ADDR"] == "127.0.0.1" and j["env"]["NCCL_DEBUG"] == "INFO"
    assert set(j["refusals"]) == {"2", "3"}


def test_failing_rank_prints_its_comm_env():
    # a rank whose device is missing says which device it wanted; the launcher's stderr carries the ranks' comm env
    p = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "5", "--warmup", "1"], capture_output=True,
                       text=True, timeout=600, env=_env())
    assert p.returncode != 0 and "comm env of the ranks" in p.stderr and "HSA_ENABLE_IPC_MODE_LEGACY" in p.stderr
