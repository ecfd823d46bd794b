"""helpers of the full-size gates (tests/test_gpu_fullsize.py, tests/test_gpu_sharded_fullsize.py)"""
import os
import threading
import time


def oracle_sample(make_engine, items, check, budget_s, at_least, threads=None):
    """`check(engine, item)` for a random sample of items against the oracle on several host threads (one oracle engine per thread; the
    calls release the GIL): as many as the time budget allows, never fewer than `at_least`.  Returns the number checked."""
    threads = threads or max(1, min(8, (os.cpu_count() or 2) // 2))
    t_end = time.time() + budget_s
    done = [0] * threads
    errors = []
    per = -(-at_least // threads)

    def work(t):
        try:
            e = make_engine()
            for j in range(t, len(items), threads):
                if time.time() > t_end and done[t] >= per:
                    break
                check(e, items[j])
                done[t] += 1
        except BaseException as ex:       # noqa: BLE001 — reported by the caller's thread
            errors.append(ex)
    ths = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    if errors:
        raise errors[0]
    assert sum(done) >= at_least, done
    return sum(done)
