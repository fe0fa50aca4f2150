"""The asynchronous, pipelined host-to-host path of the C ABI (zly_submit / zly_poll / zly_wait, include/zly.h): frames handed
over by several host threads, batched by the engine, uploaded while the previous batch computes.  Results must be the
bytes the synchronous path gives."""
import ctypes as C
import threading

import numpy as np
import pytest

import zly
import zly_model as zm
from oracle_lib import det_fields_equal

pytestmark = pytest.mark.gpu


def _frames():
    out = list(zm.synth_frames(6, 416, 416, seed=61, rects=False))
    out += [zm.synth_frames(1, 800, 600, seed=62, rects=False)[0], zm.synth_frames(1, 320, 240, seed=63)[0],
            zm.synth_frames(1, 64, 48, seed=64, rects=False)[0]]
    return [np.ascontiguousarray(f) for f in out]


def test_submit_wait_from_four_threads_equals_serial_detect_fp32(weights_path):
    """fp32 engine: one kernel configuration for every batch size, so however the engine happens to batch the 72 requests of
    four threads (mixed frame sizes in one batch, per-frame descriptors, ring slots reused several times), every ticket must
    return exactly the detections of a serial zly_detect of its frame."""
    frames = _frames()
    e = zly.Engine(weights_path, dtype=zly.DTYPE_FP32, max_batch=8, max_dets=128, warmup_runs=1)
    serial = [e.detect(f, cap=128) for f in frames]
    results, errors = {}, []

    # 4 x 18 = 72 frames, ring = 4 slots x 8: threads that only wait after submitting everything would dead-lock the ring, so a
    # dedicated consumer drains tickets as they are produced (the shape of the plugin's completion thread)
    tickets, lock, done = [], threading.Lock(), threading.Event()

    def producer(tid):
        try:
            for rep in range(2):
                for i in range(len(frames)):
                    k = (i + tid) % len(frames)
                    t = e.submit(frames[k])
                    with lock:
                        tickets.append((k, t))
        except Exception as exc:          # pragma: no cover
            errors.append(exc)

    def consumer(total):
        got = 0
        try:
            while got < total:
                with lock:
                    item = tickets.pop(0) if tickets else None
                if item is None:
                    if done.is_set() and not tickets:
                        break
                    continue
                k, t = item
                results[(got, k)] = e.wait(t, cap=128)
                got += 1
        except Exception as exc:          # pragma: no cover
            errors.append(exc)

    ths = [threading.Thread(target=producer, args=(t,)) for t in range(4)]
    cons = threading.Thread(target=consumer, args=(4 * 2 * len(frames),))
    cons.start()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    done.set()
    cons.join(timeout=120)
    assert not errors, errors
    assert len(results) == 4 * 2 * len(frames)
    for (_, k), (dets, n) in results.items():
        assert n == serial[k][1] and det_fields_equal(dets, serial[k][0]), k
        assert n == 0 or np.all(dets["timestamp"] > 1_600_000_000_000)
    st = e.stats()
    assert st["inference_errors"] == 0 and st["batches"] < len(serial) + len(results)      # requests were batched
    e.close()


def test_submit_wait_bf16_single_frame_batches_are_bit_identical_to_detect(weights_path):
    """bf16 engine with max_batch = 1: every pipelined batch is one frame, i.e. the kernels of zly_detect -- same bytes; also the
    ticket rules: poll before/after, a ticket cannot be consumed twice, a wrong byte count fails in zly_submit with 203."""
    frames = _frames()
    e = zly.Engine(weights_path, max_batch=1, max_dets=128, warmup_runs=1, flags=zly.FLAG_NO_HEAD_TENSOR | zly.FLAG_ASYNC_NMS)
    serial = [e.detect(f, cap=128) for f in frames]
    for rep in range(3):
        for f, (sd, sn) in zip(frames, serial):
            t = e.submit(f)
            dets, n = e.wait(t, cap=128)
            assert n == sn and det_fields_equal(dets, sd)
    t = e.submit(frames[0])
    while not e.poll(t):
        pass
    e.wait(t)
    with pytest.raises(zly.ZlyError) as ei:
        e.wait(t)
    assert ei.value.code == zly.ERR_INVALID_ARGUMENT
    with pytest.raises(zly.ZlyError) as ei:
        e.submit(frames[0], nbytes=frames[0].nbytes - 1)
    assert ei.value.code == zly.ERR_INVALID_INPUT and "expected 519168" in ei.value.message
    assert e.stats()["inference_errors"] == 1
    e.close()


def test_mixed_sizes_do_not_synchronise_the_device(weights_path):
    """alternating 416x416 / 800x600 requests: every call uploads a new descriptor table from the pinned ring (more calls than
    ring entries, so entries are reused); results stay those of the serial path and phase timing is sampled in production."""
    e = zly.Engine(weights_path, dtype=zly.DTYPE_FP32, max_batch=2, max_dets=128, warmup_runs=1)
    a = zm.synth_frames(1, 416, 416, seed=71, rects=False)[0]
    b = zm.synth_frames(1, 800, 600, seed=72, rects=False)[0]
    ra, rb = e.detect(a, cap=128), e.detect(b, cap=128)
    for i in range(40):
        d, n = e.detect(a if i % 2 == 0 else b, cap=128)
        want = ra if i % 2 == 0 else rb
        assert n == want[1] and det_fields_equal(d, want[0])
    res = e.detect_batch([b, a], cap=128)
    assert res[0][1] == rb[1] and res[1][1] == ra[1] and det_fields_equal(res[0][0], rb[0]) and det_fields_equal(res[1][0], ra[0])
    st = e.stats()
    assert st["sampled_frames"] >= 2 and st["sampled_preprocess_ms"] > 0 and st["sampled_forward_ms"] > 0 and st["sampled_postprocess_ms"] > 0
    e.close()
