"""GPU (-m gpu): the stream mode (mi355ndt_stream_*; include/mi355_ndt.h) -- batches that arrive one after the other, as the reference's node
consumes a stream of frames (scan_matching_odom_nodelet.cpp:144-183) -- against the synchronous batch calls: a pair's result must not depend
on the launch that happened to serve it, so EVERY result word of every batch equals mi355ndt_batch_align's (whose bits the other GPU
tests hold against the oracle).  Also the safety net of the one-launch align: a launch that gives up is re-run by the round-based path."""
import threading
import time

import numpy as np
import pytest

from conftest import se3_err
from lv_slam_amd import ndt, synth
from oracle import oracle_py as O

pytestmark = pytest.mark.gpu

WORDS = ("final", "score", "trans_probability", "iterations", "converged", "sweeps", "status", "hits_last")


def same(a, b):
    return all(np.array_equal(a[w], b[w]) if w == "final" else a[w] == b[w] for w in WORDS)


def make_batches(first_id, sizes, naz, ragged=True):
    """distinct batches of resident pairs: (T, S, counts, guesses[n,4,4]) per batch"""
    import torch
    dev = torch.device("cuda:0")
    n = naz * 64
    out, pid = [], first_id
    for bi, B in enumerate(sizes):
        T = torch.empty(B, 3, n, device=dev)
        S = torch.empty(B, 3, n, device=dev)
        for k in range(B):
            t, s, _ = synth.make_pair(pid, naz, device=dev)
            T[k] = t.T
            S[k] = s.T
            pid += 1
        cnt = [n - (613 * ((k + bi) % 5) if ragged else 0) for k in range(B)]
        G = np.stack([synth.default_guess() for _ in range(B)])
        G[bi % 3::3, 0, 3] += 0.35                      # uneven iteration counts inside every batch
        G[1::7, 1, 3] -= 0.2
        out.append((T, S, cnt, G.astype(np.float32)))
    torch.cuda.synchronize()
    return out, n


def colmajor(G):
    return np.ascontiguousarray(np.transpose(G, (0, 2, 1))).reshape(len(G), 16)


def sync_results(batches, n, kw, async_opt=0):
    eng = ndt.Engine(ndt.default_params(**kw))
    eng.set_option(ndt.OPT_ASYNC_ALIGN, async_opt)
    ref = []
    for T, S, cnt, G in batches:
        eng.batch_bind_device(T.data_ptr(), [n] * len(cnt), n, S.data_ptr(), cnt, n)
        eng.batch_build_targets()
        ref.append(eng.batch_align(G))
    eng.close()
    return ref


def stream_results(batches, n, kw, nctx, thresh, depth=None, opts=()):
    """submit the batches with at most `depth` (default: all the contexts allow) uncollected, collect in order"""
    eng = ndt.Engine(ndt.default_params(**kw))
    eng.set_option(ndt.OPT_STREAM_THRESHOLD, thresh)
    for o, v in opts:
        eng.set_option(o, v)
    eng.profile_enable(True)
    eng.stream_begin(nctx, max(len(b[2]) for b in batches), n, n)
    depth = nctx if depth is None else depth
    ids, got = [], []
    for k, (T, S, cnt, G) in enumerate(batches):
        if len(ids) - len(got) >= depth:
            got.append(eng.stream_collect(ids[len(got)], len(batches[len(got)][2])))
        ids.append(eng.stream_submit(T.data_ptr(), [n] * len(cnt), n, S.data_ptr(), cnt, n, colmajor(G)))
    while len(got) < len(ids):
        got.append(eng.stream_collect(ids[len(got)], len(batches[len(got)][2])))
    assert ids == list(range(len(batches)))
    pr = eng.profile_get()
    eng.stream_end()
    eng.close()
    return got, pr


@pytest.mark.parametrize("mode,variant,nctx,thresh,reserve", [(ndt.DIRECT7, 0, 3, -1, 0), (ndt.DIRECT7, 0, 2, 8, 0), (ndt.DIRECT1, 1, 4, 24, 0), (ndt.DIRECT1, 1, 2, 0, 0),
                                                             (ndt.DIRECT7, 1, 3, 128, 0), (ndt.KDTREE, 0, 3, 6, 0),
                                                             (ndt.DIRECT7, 0, 3, -1, 64), (ndt.DIRECT1, 1, 4, 24, 128), (ndt.DIRECT7, 1, 3, 128, 32)])
def test_stream_equals_the_synchronous_batches(mode, variant, nctx, thresh, reserve):
    """Seven distinct ragged batches (17..40 pairs of 32,768 points, uneven iteration counts) streamed with 2, 3 or 4 resident contexts and
    hand-over thresholds from "never" to "always": every result word of every pair equals the round-based synchronous align's -- whether
    the pair finished in its own launch, or was suspended once or several times and finished under later batches."""
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=mode, variant=variant)
    batches, n = make_batches(2000, [40, 33, 40, 17, 40, 29, 40], 512)
    ref = sync_results(batches, n, kw)
    got, pr = stream_results(batches, n, kw, nctx, thresh, opts=((ndt.OPT_STREAM_RESERVE, reserve),))      # reserve > 0: every batch's build runs beside the previous launch
    for bi, (r, g) in enumerate(zip(ref, got)):
        assert len(r) == len(g)
        for k, (x, y) in enumerate(zip(r, g)):
            assert same(x, y), (bi, k, x, y)
    assert pr["stream_redone"] == 0 and pr["async_fallbacks"] == 0
    assert pr["stream_launches"] >= len(batches)            # one per batch (+ the flush of the last stragglers)
    if thresh == 0:
        assert pr["stream_carried"] == 0 and pr["stream_launches"] == len(batches)
    elif thresh >= 8 or thresh < 0:
        assert pr["stream_carried"] > 0                     # pairs really did travel between launches
    assert len({r["iterations"] for r in ref[0]}) >= 3


def test_stream_against_the_oracle_and_collect_order():
    """The stream's results against the oracle directly (sample of pairs, the bars of the batch tests), with the caller collecting as late as
    the contexts allow and one batch collected before its successor is even submitted (its stragglers are then flushed by collect itself)."""
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
    batches, n = make_batches(2400, [24, 24, 24, 24], 512)
    eng = ndt.Engine(ndt.default_params(**kw))
    eng.stream_begin(3, 24, n, n)
    sub = lambda b: eng.stream_submit(b[0].data_ptr(), [n] * 24, n, b[1].data_ptr(), b[2], n, colmajor(b[3]))
    i0 = sub(batches[0])
    r0 = eng.stream_collect(i0, 24)                          # nothing newer: flushed
    i1, i2 = sub(batches[1]), sub(batches[2])
    i3 = sub(batches[3])                                     # three uncollected = the three contexts
    with pytest.raises(ndt.NDTError):                        # a fourth needs batch 1's context
        sub(batches[0])
    r1, r2, r3 = eng.stream_collect(i1, 24), eng.stream_collect(i2, 24), eng.stream_collect(i3, 24)
    with pytest.raises(ndt.NDTError):
        eng.stream_collect(i1, 24)                           # collected already
    with pytest.raises(ndt.NDTError):
        eng.batch_build_targets()                            # the handle belongs to the stream until stream_end
    eng.stream_end()
    eng.close()
    op = O.default_params(**kw)
    import torch
    for (T, S, cnt, G), res in zip(batches, (r0, r1, r2, r3)):
        for k in (0, 9, 23):
            tgt = T[k].T.contiguous().cpu().numpy()
            src = S[k, :, :cnt[k]].T.contiguous().cpu().numpy()
            ro = O.align(O.Grid(tgt, op), src, G[k])
            assert ro["iterations"] == res[k]["iterations"] and ro["converged"] == res[k]["converged"] and ro["hits_last"] == res[k]["hits_last"]
            dt, dr = se3_err(ro["final"], res[k]["final"])
            assert dt < 1e-4 and dr < 1e-5
    torch.cuda.synchronize()


def test_stream_batch_that_exceeds_the_build_plan_is_rerun():
    """Builds inside the stream do not wait for the grids' sizes: they use a plan made by the first build.  A batch whose targets are three
    times as wide (27 x the cells) does not fit it: the device withholds its grids, collect re-runs it synchronously, and the plan grows --
    results equal the synchronous path's, the batches around it are untouched."""
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
    batches, n = make_batches(2600, [20, 20, 20, 20], 256, ragged=False)
    T, S, cnt, G = batches[1]
    batches[1] = (T * 3.0, S * 3.0, cnt, G)                  # a much larger scene
    T, S, cnt, G = batches[2]
    batches[2] = (T * 3.0, S * 3.0, cnt, G)                  # ... and the next one fits the new plan
    ref = sync_results(batches, n, kw)
    got, pr = stream_results(batches, n, kw, 3, -1)
    for bi, (r, g) in enumerate(zip(ref, got)):
        for k, (x, y) in enumerate(zip(r, g)):
            assert same(x, y), (bi, k, x, y)
    assert 1 <= pr["stream_redone"] <= 2                     # (batch 2 was submitted before batch 1's collect re-made the plan)


def test_one_launch_align_that_gives_up_is_rerun_by_the_rounds():
    """MI355NDT_OPT_DEBUG_ASYNC_ABORT makes one wave of the persistent launch give up the way a wave whose ticket never came does
    (bounded polls, ndt_async.hpp).  The caller must not see it: batch_align re-runs the batch through the round-based path -- same bits --
    and counts the event; the stream mode re-runs every unfinished batch."""
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
    batches, n = make_batches(2800, [40, 40, 40], 512)
    ref = sync_results(batches, n, kw)
    eng = ndt.Engine(ndt.default_params(**kw))
    eng.set_option(ndt.OPT_ASYNC_ALIGN, 2)                   # the one-launch align whatever the batch size
    eng.profile_enable(True)
    T, S, cnt, G = batches[0]
    eng.batch_bind_device(T.data_ptr(), [n] * 40, n, S.data_ptr(), cnt, n)
    eng.batch_build_targets()
    for pos, fallbacks in ((-1, 0), (3, 1), (700, 2), (-1, 2)):
        eng.set_option(ndt.OPT_DEBUG_ASYNC_ABORT, pos)
        res = eng.batch_align(G)
        for k, (x, y) in enumerate(zip(ref[0], res)):
            assert same(x, y), (pos, k)
        assert eng.profile_get()["async_fallbacks"] == fallbacks
    eng.close()
    got, pr = stream_results(batches, n, kw, 3, -1, opts=((ndt.OPT_DEBUG_ASYNC_ABORT, 40),))
    for bi, (r, g) in enumerate(zip(ref, got)):
        for k, (x, y) in enumerate(zip(r, g)):
            assert same(x, y), (bi, k)
    assert pr["async_fallbacks"] >= 1 and pr["stream_redone"] >= 1


def test_one_aborted_launch_is_recovered_once():
    """ONE launch of a stream gives up (the hook is armed for the third submit only) while earlier batches still have pairs riding in it and later batches
    follow: the recovery runs once -- one fallback counted, only the batches that were unfinished at that moment re-run, the batches submitted afterwards
    stream normally -- and every result word still equals the synchronous align's.  (A recovery per collect that walked past the aborted launch's status
    slot used to mark healthy later batches for a synchronous re-run as well.)"""
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
    batches, n = make_batches(3100, [40, 40, 40, 40, 40, 40, 40], 512)
    ref = sync_results(batches, n, kw)
    eng = ndt.Engine(ndt.default_params(**kw))
    eng.set_option(ndt.OPT_STREAM_THRESHOLD, 16)             # pairs of batches 0 and 1 are carried into the launches behind them
    eng.profile_enable(True)
    eng.stream_begin(4, 40, n, n)
    ids, got = [], []
    for k, (T, S, cnt, G) in enumerate(batches):
        if len(ids) - len(got) >= 4:
            got.append(eng.stream_collect(ids[len(got)], 40))
        eng.set_option(ndt.OPT_DEBUG_ASYNC_ABORT, 40 if k == 2 else -1)
        ids.append(eng.stream_submit(T.data_ptr(), [n] * 40, n, S.data_ptr(), cnt, n, colmajor(G)))
    while len(got) < len(ids):
        got.append(eng.stream_collect(ids[len(got)], 40))
    pr = eng.profile_get()
    eng.stream_end()
    eng.close()
    for bi, (r, g) in enumerate(zip(ref, got)):
        for k, (x, y) in enumerate(zip(r, g)):
            assert same(x, y), (bi, k)
    assert pr["async_fallbacks"] == 1, pr
    assert 1 <= pr["stream_redone"] <= 4, pr                 # at most the batches resident when the launch gave up


@pytest.mark.parametrize("mask", [0x0F, 0xA5, 0x01])
def test_rings_without_workgroups_of_their_own_are_served_by_the_others(mask):
    """Tickets go round the eight rings and a ring is served by the workgroups of one XCD; nothing guarantees that every XCD holds workgroups
    of a launch (two engines launching at once can split the XCDs between them: seen once in ~10^3 launches as a launch that gave up after
    its poll budget).  MI355NDT_OPT_DEBUG_ASYNC_RINGS takes the workgroups of some rings out of the launch on purpose: the waves that are
    there must serve every ring's published positions (ndt_async.hpp: a waiting wave takes servable positions of other rings), the launch
    must end with every pair finished -- ONE launch, no fallback -- and with the same bits; also through the stream."""
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
    batches, n = make_batches(2900, [40, 40, 40], 512)
    ref = sync_results(batches, n, kw)
    eng = ndt.Engine(ndt.default_params(**kw))
    eng.set_option(ndt.OPT_ASYNC_ALIGN, 2)                   # the one-launch align whatever the batch size
    eng.set_option(ndt.OPT_DEBUG_ASYNC_RINGS, mask)
    eng.profile_enable(True)
    T, S, cnt, G = batches[0]
    eng.batch_bind_device(T.data_ptr(), [n] * 40, n, S.data_ptr(), cnt, n)
    eng.batch_build_targets()
    t0 = time.time()
    for rep in range(3):
        res = eng.batch_align(G)
        for k, (x, y) in enumerate(zip(ref[0], res)):
            assert same(x, y), (rep, k)
    pr = eng.profile_get()
    assert pr["async_fallbacks"] == 0 and pr["sweep_launches"] == 3 and pr["update_launches"] == 0
    assert time.time() - t0 < 5.0                            # (a launch that waits for its poll budget takes ~10 s)
    eng.close()
    got, pr = stream_results(batches, n, kw, 3, -1, opts=((ndt.OPT_DEBUG_ASYNC_RINGS, mask),))
    for bi, (r, g) in enumerate(zip(ref, got)):
        for k, (x, y) in enumerate(zip(r, g)):
            assert same(x, y), (bi, k)
    assert pr["async_fallbacks"] == 0 and pr["stream_redone"] == 0


def test_two_engines_on_two_threads_share_the_gpu():
    """The one-launch align sizes its grid to be resident as a whole and its waves wait for each other's tickets; a second engine on the same
    GPU (another rank's, another thread's) competes for the same CUs.  Residency is no condition of correctness -- positions are claimed,
    the resident waves of a ring do all of its work, late workgroups join or find the launch over -- and this holds it: two engines on two
    host threads, both above the big-batch threshold (18 x 65,536 points = 2,304 work items > 2,048 resident waves), 200 aligns each at the
    same time: every result equal to the first, and not one launch gave up."""
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
    batches, n = make_batches(3000, [18, 18], 1024, ragged=False)
    out, err = [None, None], []

    def drive(i):
        try:
            T, S, cnt, G = batches[i]
            eng = ndt.Engine(ndt.default_params(**kw))
            eng.profile_enable(True)
            eng.batch_bind_device(T.data_ptr(), [n] * 18, n, S.data_ptr(), cnt, n)
            eng.batch_build_targets()
            gc = colmajor(G)
            first = eng.batch_align(G)
            res = (ndt.Result * 18)()
            bad = 0
            for _ in range(200):
                eng.batch_align_raw(gc, res)
                bad += sum(not (np.array_equal(np.array(r.final_colmajor, np.float32).reshape(4, 4).T, f["final"]) and r.score == f["score"] and r.iterations == f["iterations"])
                           for r, f in zip(res, first))
            pr = eng.profile_get()
            eng.close()
            out[i] = (first, bad, pr)
        except Exception as e:                              # noqa: BLE001
            err.append(e)

    th = [threading.Thread(target=drive, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not err, err
    ref = sync_results(batches, n, kw)
    for i in range(2):
        first, bad, pr = out[i]
        assert bad == 0
        assert pr["async_fallbacks"] == 0
        assert pr["sweep_launches"] == 201 and pr["update_launches"] == 0      # every align was ONE launch
        for x, y in zip(ref[i], first):
            assert same(x, y)


def test_cross_xcd_message_passing_litmus():
    """The hand-over protocol of the one-launch align (sc1 stores -> s_waitcnt vmcnt(0) -> sc1 announcement; sc1 poll -> sc1 loads) as a
    litmus kernel of its own: 256 writer / reader pairs three XCDs apart, 10.2 million hand-overs of a 44-word row, no stale word, no timeout."""
    import os, subprocess
    import __graft_entry__ as entry
    exe = entry.build_mp_litmus() if os.path.exists(entry.HIPCC) else os.path.join(entry.ROOT, "tests", "hip", "mp_litmus")
    r = subprocess.run([exe, "40000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout, r.stderr)
    kv = dict(f.split("=") for f in r.stdout.split())
    assert int(kv["handovers"]) >= 10_000_000 and int(kv["errors"]) == 0 and int(kv["timeouts"]) == 0, r.stdout


def test_stream_pose_records_are_written_by_the_device():
    """mi355ndt_stream_pose_records: a batch's 96-byte gather records appear in the caller's device block as its pairs finish --
    written inside the persistent launch by the wave that finalises the pair -- and equal what the host packs from the collected results;
    rows beyond the batch are padding (pair_id = -1)."""
    import ctypes as C
    import torch
    from lv_slam_amd import dist as shard
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT7, variant=0)
    batches, n = make_batches(3300, [30, 22, 30], 512)
    eng = ndt.Engine(ndt.default_params(**kw))
    eng.stream_begin(3, 30, n, n)
    cap = 32
    blocks = [torch.full((cap, shard.REC_WORDS), 7, device="cuda:0", dtype=torch.int32) for _ in batches]
    ids = []
    for k, (T, S, cnt, G) in enumerate(batches):
        eng.stream_pose_records(blocks[k].data_ptr(), cap, 3 + k, 8)          # pair_id = 3 + k + 8 b
        ids.append(eng.stream_submit(T.data_ptr(), [n] * len(cnt), n, S.data_ptr(), cnt, n, colmajor(G)))
    lib = ndt.load_library()
    for k, (T, S, cnt, G) in enumerate(batches):
        B = len(cnt)
        res = (ndt.Result * B)()
        eng.stream_collect_raw(ids[k], res)
        want = np.zeros((cap, shard.REC_WORDS), np.int32)
        assert lib.mi355ndt_pack_pose_records(C.cast(res, C.c_void_p), B, 3 + k, 8, want.ctypes.data_as(C.c_void_p), cap) == 0
        got = blocks[k].cpu().numpy()
        assert np.array_equal(got, want), k
        rec = shard.unpack_records(blocks[k])
        assert sorted(rec) == [3 + k + 8 * b for b in range(B)] and rec[3 + k]["iterations"] == res[0].iterations
    eng.stream_end()
    eng.close()


@pytest.mark.parametrize("rec_floats,nctx", [(8, 3), (4, 2)])
def test_stream_submit_host_equals_the_synchronous_host_path(rec_floats, nctx):
    """HOST clouds through the stream (mi355ndt_stream_submit_host: staged into the batch context's pinned slots, over PCIe under the launches of earlier
    batches) against the same clouds through the synchronous drop-in calls (batch_set_clouds + build + align) and against the device-resident synchronous
    align: every result word equal.  Ragged clouds, 32-byte pcl::PointXYZI and 16-byte pcl::PointXYZ records, five batches through two / three contexts."""
    kw = dict(resolution=1.0, trans_epsilon=0.01, max_iterations=64, neighbor_mode=ndt.DIRECT1, variant=1)
    batches, n = make_batches(3300, [24, 17, 24, 9, 24], 256)
    ref = sync_results(batches, n, kw)
    stride = rec_floats * 4
    host = []
    for T, S, cnt, G in batches:
        B = len(cnt)
        tg = np.zeros((B, n, rec_floats), np.float32); sr = np.zeros((B, n, rec_floats), np.float32)
        tg[:, :, :3] = T.permute(0, 2, 1).cpu().numpy(); sr[:, :, :3] = S.permute(0, 2, 1).cpu().numpy()
        tg[:, :, 3] = 1.0; sr[:, :, 3] = 1.0
        host.append((tg, sr))
    ptrs = lambda a: np.uint64(a.ctypes.data) + np.arange(a.shape[0], dtype=np.uint64) * np.uint64(a.shape[1] * stride)
    # the synchronous drop-in path
    eng = ndt.Engine(ndt.default_params(**kw))
    for (T, S, cnt, G), (tg, sr), r in zip(batches, host, ref):
        B = len(cnt)
        eng.batch_reserve(B, n, n)
        eng.batch_set_clouds_raw(0, ptrs(tg), np.full(B, n, np.uint64), ptrs(sr), np.asarray(cnt, np.uint64), stride, 4)
        eng.batch_build_targets()
        got = eng.batch_align(G)
        for k, (x, y) in enumerate(zip(r, got)):
            assert same(x, y), k
    # ... and streamed
    eng.profile_reset()
    eng.stream_begin(nctx, 24, n, n)
    ids, got = [], []
    for bi, ((T, S, cnt, G), (tg, sr)) in enumerate(zip(batches, host)):
        if len(ids) - len(got) >= nctx:
            got.append(eng.stream_collect(ids[len(got)], len(batches[len(got)][2])))
        B = len(cnt)
        ids.append(eng.stream_submit_host_raw(ptrs(tg), np.full(B, n, np.uint64), ptrs(sr), np.asarray(cnt, np.uint64), stride, colmajor(G), 4))
        tg[:] = np.nan; sr[:] = np.nan                   # the call has returned: the caller's memory is the caller's again
    while len(got) < len(ids):
        got.append(eng.stream_collect(ids[len(got)], len(batches[len(got)][2])))
    pr = eng.profile_get()
    # a batch whose counts exceed what stream_begin was told is refused, not truncated
    with pytest.raises(ndt.NDTError):
        big = np.zeros((1, 2 * n, rec_floats), np.float32)
        eng.stream_submit_host_raw(ptrs(big), np.array([2 * n], np.uint64), ptrs(big), np.array([n], np.uint64), stride, colmajor(batches[0][3][:1]), 2)
    eng.stream_end()
    eng.close()
    for bi, (r, g) in enumerate(zip(ref, got)):
        assert len(r) == len(g)
        for k, (x, y) in enumerate(zip(r, g)):
            assert same(x, y), (bi, k)
    assert pr["cloud_uploads"] == 2 * sum(len(b[2]) for b in batches) and pr["async_fallbacks"] == 0


def test_stream_reserve_defaults_follow_the_search_and_the_batch_size():
    """MI355NDT_OPT_STREAM_RESERVE left at its default: the next batch's build runs beside the launch (profile.stream_reserved_slots > 0) for DIRECT1
    (128 workgroup slots, 96 for clouds beyond 98,304 points) and for DIRECT7 with batches of up to 768 x 65,536 target points (ndt_omp 64 = eight per XCD,
    ndt_pca 32), and between the launches (0) for large DIRECT7 batches, with fewer than three contexts and for every other search; an explicit value wins.  (No result bit
    depends on it: test_stream_equals_the_synchronous_batches, and bench.py compares every streamed batch with the synchronous results.)"""
    def slots(kw, nctx, max_pairs, pts, opts=()):
        eng = ndt.Engine(ndt.default_params(resolution=1.0, trans_epsilon=0.01, max_iterations=64, **kw))
        for o, v in opts:
            eng.set_option(o, v)
        eng.stream_begin(nctx, max_pairs, pts, pts)
        p = eng.profile_get()
        eng.stream_end()
        eng.close()
        return p["stream_reserved_slots"], p["stream_launch_slots"]
    omp7, pca7, pca1 = dict(neighbor_mode=ndt.DIRECT7, variant=0), dict(neighbor_mode=ndt.DIRECT7, variant=1), dict(neighbor_mode=ndt.DIRECT1, variant=1)
    r, launch = slots(omp7, 3, 271, 65536)
    assert r == 64 and launch > 0 and (launch + r) % 8 == 0
    all_slots = launch + r
    assert slots(omp7, 2, 271, 65536) == (0, all_slots)                       # two contexts: the build has nowhere to run ahead
    assert slots(omp7, 3, 768, 65536)[0] == 64 and slots(omp7, 3, 769, 65536)[0] == 0
    assert slots(omp7, 4, 128, 131072)[0] == 64
    assert slots(pca7, 3, 128, 131072) == (32, all_slots - 32) and slots(pca7, 3, 1536, 65536)[0] == 0
    assert slots(pca1, 3, 271, 65536) == (128, all_slots - 128) and slots(pca1, 3, 1536, 65536)[0] == 128 and slots(pca1, 3, 128, 131072)[0] == 96
    assert slots(dict(neighbor_mode=ndt.KDTREE, variant=0), 3, 64, 65536)[0] == 0
    assert slots(omp7, 3, 271, 65536, opts=((ndt.OPT_STREAM_RESERVE, 32),)) == (32, all_slots - 32)
    assert slots(pca1, 3, 271, 65536, opts=((ndt.OPT_STREAM_RESERVE, 0),)) == (0, all_slots)
    rt, lt = slots(omp7, 3, 271, 65536, opts=((ndt.OPT_ARITH, 1),))          # tolerance arithmetic: twice the workgroups per CU, the same 64
    assert rt == 64 and lt + rt == 2 * all_slots
