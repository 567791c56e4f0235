"""Stress of the library-free transport between PROCESSES sharing one GPU: the 8-process 4 x 2 replay of tests/test_gpu_brick.py
(test_replayed_cycles_between_processes_with_the_peer_transport) N times in a row; prints every run's outcome.
    python tools/stress_peer.py 40"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import test_gpu_brick as t
    fails = 0
    for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        port = t._free_port()
        procs = [ctx.Process(target=t._peer_replay_worker, args=(r, 8, port, q, (4, 2, 1), (16, 10, 5), 2)) for r in range(8)]
        t0 = time.time()
        for p in procs:
            p.start()
        try:
            res = [q.get(timeout=150) for _ in procs]
        except Exception as e:  # noqa: BLE001
            res = [(-1, "no answer: %r" % (e,))]
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
        bad = [(r[0], str(r[1]).strip().splitlines()[-1][:160]) for r in res if r[1] != "ok"]
        print(i, "%.1fs" % (time.time() - t0), "ok" if not bad else bad, flush=True)
        fails += bool(bad)
    print("failures", fails)


if __name__ == "__main__":
    main()
