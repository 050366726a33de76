"""`bench.py --gpus N` started PLAINLY (no launcher): the parent spawns one rank per GPU before anything touches a GPU,
the ranks meet over the product's stdlib TCP group, rank 0's JSON line is relayed and a failing rank fails the job.
No GPU and no torch in these processes (`--rendezvous-only` stops after the start-up)."""
import json
import os
import socket
import subprocess
import sys
import threading

import pytest

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                                              "HIPEIG_RDZV_PORT", "TORCHELASTIC_RUN_ID")}
    return env


@pytest.mark.timeout(180)
@pytest.mark.parametrize("n", [2, 3, 8])                 # 8: the command line of a SCALE run, `python bench.py --gpus 8`
def test_plain_start_spawns_the_ranks_and_relays_rank0(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--rendezvous-only"], capture_output=True, text=True,
                       env=_clean_env(), timeout=150)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE JSON line on stdout, whatever the children print
    out = json.loads(lines[0])
    assert out == {"rendezvous": "ok", "n_ranks_seen": n, "n_gpus": n, "launcher": "bench.py"}


@pytest.mark.timeout(120)
def test_a_launchers_environment_is_honoured_and_checked():
    """WORLD_SIZE set by a launcher: no self-spawn; a world size that disagrees with --gpus is an error before any GPU call."""
    env = _clean_env()
    env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="3", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rendezvous-only"], capture_output=True, text=True, env=env,
                       timeout=100)
    assert r.returncode != 0 and "must agree" in r.stderr and r.stdout.strip() == ""


@pytest.mark.timeout(120)
def test_launch_local_reports_a_failing_rank_and_stops_the_rest(tmp_path):
    from eigensolvers_amd.distributed import launch_local
    prog = tmp_path / "rank.py"
    prog.write_text("import os, sys, time\n"
                    "r = int(os.environ['RANK'])\n"
                    "assert os.environ['WORLD_SIZE'] == '3' and os.environ['LOCAL_RANK'] == str(r)\n"
                    "if r == 1:\n    sys.exit(7)\n"
                    "if r == 0:\n    print('{\"rank0\": true}', flush=True)\n"
                    "time.sleep(60)\n")
    rc, out = launch_local([str(prog)], 3, timeout=50)
    assert rc == 7 and '"rank0"' in out                      # rank 1's status; ranks 0 and 2 were terminated, not waited for


@pytest.mark.timeout(120)
def test_tcp_group_survives_strangers_and_truncated_greetings(monkeypatch):
    """a garbled or partial hello must not kill rank 0; a peer counts as joined only after its ACK."""
    from eigensolvers_amd import distributed as D
    port = D.free_port()
    monkeypatch.setenv("HIPEIG_RDZV_PORT", str(port))
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    res = {}

    def rank0():
        g = D.TcpGroup(0, 2, timeout=60.0)
        res["g0"] = g.allgather(b"zero")
        res["b0"] = g.bcast(b"id" * 64)
        g.close()

    t0 = threading.Thread(target=rank0)
    t0.start()
    import time
    for attempt in range(200):                               # wait for the listener
        try:
            s = socket.create_connection(("127.0.0.1", port), timeout=1.0)
            break
        except OSError:
            time.sleep(0.05)
    s.sendall(b"\x05\x00\x00")                               # a truncated length prefix, then silence
    s.close()
    s = socket.create_connection(("127.0.0.1", port), timeout=1.0)
    s.sendall((9).to_bytes(8, "little") + b"GET / HTT")       # a stranger
    s.close()
    s = socket.create_connection(("127.0.0.1", port), timeout=1.0)
    hello = D._MAGIC + b"|" + D._run_token() + b"|notanumber"
    s.sendall(len(hello).to_bytes(8, "little") + hello)        # right prefix, garbled rank
    s.close()
    g1 = D.TcpGroup(1, 2, timeout=60.0)
    assert g1.allgather(b"one") == [b"zero", b"one"]
    assert g1.bcast(b"") == b"id" * 64
    g1.close()
    t0.join(30)
    assert res["g0"] == [b"zero", b"one"] and res["b0"] == b"id" * 64
