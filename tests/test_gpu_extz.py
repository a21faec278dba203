"""Parity of extz_kernel (through moni_extz_batch) with the ksw2 restatement: every ksw_extz_t field and
every CIGAR, bit-exact, over mixed problem sizes, all flag combinations the reference uses, wildcards and
empty inputs; plus the reference's own worked example."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ["max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "score", "reach_end", "zdropped", "n_cigar"]


@pytest.fixture(scope="module", params=["registers", "lds"])
def ctx(small_case, request):
    """Both forms of the DP: register-blocked extz_kernel<NCH> and the LDS-tiled one align_kernel runs (MONI_EXTZ_LDS=1,
    read when the context is created)."""
    from moni_align_amd import capi
    idx = capi.Index(fi=small_case.fi)
    old = os.environ.get("MONI_EXTZ_LDS")
    os.environ["MONI_EXTZ_LDS"] = "1" if request.param == "lds" else "0"
    try:
        c = capi.Ctx(idx)
    finally:
        if old is None:
            del os.environ["MONI_EXTZ_LDS"]
        else:
            os.environ["MONI_EXTZ_LDS"] = old
    yield c
    c.close()
    idx.close()


def make_tasks(rng, n, qmax, tmax, flags):
    from moni_align_amd import capi
    qs, ts = [], []
    tasks = np.zeros(n, dtype=capi.DP_TASK_DTYPE)
    qo = to = 0
    for k in range(n):
        m, l = int(rng.integers(0 if k % 97 == 0 else 1, qmax + 1)), int(rng.integers(0 if k % 89 == 0 else 1, tmax + 1))
        t = rng.integers(0, 4, size=l).astype(np.uint8)
        if rng.random() < 0.8 and l > 0 and m > 0:
            q = np.resize(t, m).copy()
            mut = rng.random(m) < 0.08
            q[mut] = rng.integers(0, 4, size=int(mut.sum()))
            if m > 10 and rng.random() < 0.5:
                a = int(rng.integers(1, m - 5)); g = int(rng.integers(1, 5))
                q = np.concatenate([q[:a], q[a + g:], rng.integers(0, 4, size=g).astype(np.uint8)])
            if m > 10 and rng.random() < 0.3:
                a = int(rng.integers(1, m - 5)); g = int(rng.integers(1, 4))
                q = np.concatenate([q[:a], rng.integers(0, 4, size=g).astype(np.uint8), q[a:m - g]])
        else:
            q = rng.integers(0, 4, size=m).astype(np.uint8)
        if m and rng.random() < 0.15:
            q[int(rng.integers(0, m))] = 4
        if l and rng.random() < 0.15:
            t[int(rng.integers(0, l))] = 4
        tasks[k] = (qo, to, len(q), len(t), flags[k % len(flags)], 0)
        qs.append(q); ts.append(t)
        qo += len(q); to += len(t)
    return np.concatenate(qs), np.concatenate(ts), tasks


def check(ctx, qseq, tseq, tasks):
    from oracle import orc
    res, pool = ctx.extz_batch(qseq, tseq, tasks)
    for k in range(len(tasks)):
        t = tasks[k]
        q = qseq[int(t["q_off"]):int(t["q_off"]) + int(t["qlen"])]
        tg = tseq[int(t["t_off"]):int(t["t_off"]) + int(t["tlen"])]
        w = orc.extz(q, tg, int(t["flag"]))
        for f in FIELDS:
            assert int(res[f][k]) == w[f], (k, f, int(res[f][k]), w[f], int(t["qlen"]), int(t["tlen"]), int(t["flag"]))
        got = pool[int(res["cigar_off"][k]):int(res["cigar_off"][k]) + int(res["n_cigar"][k])]
        assert np.array_equal(got, w["cigar"]), k


def test_extz_small_and_medium(ctx):
    from oracle import orc
    rng = np.random.default_rng(3)
    flags = [orc.FLAG_SCORE_ONLY, orc.FLAG_EXTZ_ONLY | orc.FLAG_RIGHT, orc.FLAG_RIGHT, 0, orc.FLAG_EXTZ_ONLY]
    q, t, tasks = make_tasks(rng, 1500, 130, 100, flags)
    check(ctx, q, t, tasks)


def test_extz_multi_chunk(ctx):
    from oracle import orc
    rng = np.random.default_rng(4)
    flags = [orc.FLAG_SCORE_ONLY, orc.FLAG_EXTZ_ONLY | orc.FLAG_RIGHT, orc.FLAG_RIGHT]
    q, t, tasks = make_tasks(rng, 300, 260, 380, flags)
    check(ctx, q, t, tasks)
    q, t, tasks = make_tasks(rng, 60, 600, 512, flags)
    check(ctx, q, t, tasks)


def test_extz_oversize_problems(ctx):
    """Beyond the one-wave kernels' 512 target rows (2048 / 512 query bases): extz_big_kernel, state rows in HBM."""
    from oracle import orc
    rng = np.random.default_rng(5)
    flags = [orc.FLAG_SCORE_ONLY, orc.FLAG_EXTZ_ONLY | orc.FLAG_RIGHT, orc.FLAG_RIGHT, 0]
    q, t, tasks = make_tasks(rng, 16, 1300, 1500, flags)
    assert (tasks["tlen"] > 512).sum() >= 4
    check(ctx, q, t, tasks)
    q, t, tasks = make_tasks(rng, 3, 2600, 700, flags)      # query beyond 2048 as well
    check(ctx, q, t, tasks)


def test_extz_reference_comment_example(ctx):
    from moni_align_amd import capi
    from oracle import orc
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ksw2_comment_example.json")))
    nt = lambda s: np.array([int(c) for c in s], dtype=np.uint8)
    qs = [nt(g["left"]["query"][::-1]), nt(g["right"]["query"]), nt(g["global"]["query"])]
    ts = [nt(g["left"]["target"][::-1]), nt(g["right"]["target"]), nt(g["global"]["target"])]
    tasks = np.zeros(3, dtype=capi.DP_TASK_DTYPE)
    qo = to = 0
    for k in range(3):
        tasks[k] = (qo, to, len(qs[k]), len(ts[k]), (orc.FLAG_EXTZ_ONLY | orc.FLAG_RIGHT) if k < 2 else orc.FLAG_RIGHT, 0)
        qo += len(qs[k]); to += len(ts[k])
    res, pool = ctx.extz_batch(np.concatenate(qs), np.concatenate(ts), tasks)
    assert int(res["mqe"][0]) + int(res["mqe"][1]) + 2 * g["mem_len"] == g["old_score"]
    assert int(res["score"][2]) == g["new_score"]
    c = pool[int(res["cigar_off"][2]):int(res["cigar_off"][2]) + int(res["n_cigar"][2])]
    assert "".join("%d%s" % (x >> 4, "MID"[x & 15]) for x in c) == g["global"]["cigar"]
