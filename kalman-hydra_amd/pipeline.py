"""In-process flow -> EKF streaming.

The reference computes its optical flow in a separate executable and hands it to the tracker
through files (reference README.md:26-31; ``gen_synthetic.py:41-42`` shells out to
``./bin/optical_flow_ext``); the tracker's frame loop then reads one flow file per frame
(reference run_kalmanfilter.py:78-89).  Here both halves run in one process on one GPU:

* the frames (and masks) are read from a frame source one at a time, as the reference's loop does
  (run_kalmanfilter.py:78-89), and uploaded on a copy stream into a ring of frame slots in HBM: the
  frames of the next flow series go up while the current series runs (``resident=True`` uploads the
  whole video once instead -- same results, only useful for comparisons);
* the Brox flow of consecutive frame pairs does not depend on the filter, so it is computed ahead
  of it, in launch series of 1, 2, 4, ... up to ``flow_batch`` pairs on the flow handle's own HIP
  stream (a single pair is launch-latency bound; a series amortises the ~900 launches over its
  pairs), double buffered: while the filter works through the pairs of one series, the next series
  runs beside it.  The series are queued from a helper thread -- the C calls drop the GIL -- so the
  filter's thread does not pay for their launches;
* a frame's flow planes are handed to the filter where they are, in device memory
  (``renderer.DeviceObservation`` -> ``hm_set_observation_dev``).

``VideoStream`` mirrors reference renderer.py:739-805 on an array source (no OpenCV on this path):
``.npy`` / ``.npz`` of shape (frames, H, W) or (frames, H, W, 3), 8-bit.
"""
import ctypes
import threading
import time

import os

import numpy as np

from . import _lib
from . import brox as _brox
from .renderer import DeviceObservation


def to_gray(a):
    """cv2.cvtColor(frame, COLOR_BGR2GRAY) for 8-bit frames (reference renderer.py:752,
    src/optical_flow_ext.cpp:366-368): rint(0.114 B + 0.587 G + 0.299 R).  2-D input is returned as is."""
    a = np.asarray(a)
    if a.ndim >= 3 and a.shape[-1] == 3:
        a = np.rint(0.114 * a[..., 0] + 0.587 * a[..., 1] + 0.299 * a[..., 2]).astype(np.uint8)
    return a


def load_video(fn):
    """The whole video as gray frames (frames, H, W) uint8 -- shared by run_kalmanfilter.py and
    optical_flow_ext.py, so that the tracker sees the frames its flow was computed on.  Sources (no OpenCV on this
    path; reference renderer.py:745 opens anything cv2.VideoCapture can): a NumPy ``.npy`` / ``.npz`` array of shape
    (frames, H, W[, 3]), or a multi-page TIFF stack (``.tif`` / ``.tiff``, 8-bit gray or RGB pages, read with PIL)."""
    if str(fn).lower().endswith((".tif", ".tiff")):
        from PIL import Image, ImageSequence
        with Image.open(fn) as im:
            pages = []
            for page in ImageSequence.Iterator(im):
                if page.mode not in ("L", "RGB"):
                    raise ValueError("%s: expected 8-bit gray or RGB pages, found mode %s" % (fn, page.mode))
                a = np.asarray(page)
                pages.append(a[..., ::-1] if a.ndim == 3 else a)          # RGB -> BGR, the order cv2 hands out
        if not pages or any(p.shape != pages[0].shape for p in pages):
            raise ValueError("%s: expected pages of one size" % fn)
        a = np.stack(pages)
    else:
        a = np.load(fn)
        if hasattr(a, "files"):
            a = a[a.files[0]]
        a = np.asarray(a)
    if a.dtype != np.uint8 or a.ndim not in (3, 4) or (a.ndim == 4 and a.shape[-1] != 3):
        raise ValueError("%s: expected an 8-bit array of shape (frames, H, W[, 3])" % fn)
    return np.ascontiguousarray(to_gray(a))


def threshold_mask(gray, threshold):
    """The mask imgproc.findObjectThreshold returns (reference imgproc.py:195-197): intensity above threshold."""
    return (np.asarray(gray) > threshold).astype(np.uint8)


class VideoStream:
    """reference renderer.py:739-805 over an in-memory frame stack."""

    def __init__(self, fn, threshold):
        self.threshold = threshold
        self.frames = load_video(fn) if isinstance(fn, str) else np.ascontiguousarray(to_gray(np.asarray(fn)))
        if self.frames.shape[0] < 1:
            raise ValueError("Cannot open %s" % (fn,))
        self.pos = 0
        self.nx, self.ny = self.frames.shape[1:3]
        self.frame = self.frames[0]
        self.frame_orig = self.frame.copy()
        self.grayframe = self.frame

    def read(self, backsub=True):
        """-> (ret, frame, grayframe, mask) with the background removed, or (ret, frame, grayframe)."""
        if self.pos + 1 >= self.frames.shape[0]:
            self.pos = self.frames.shape[0]
            return False, None, None, None
        self.pos += 1
        self.frame = self.grayframe = self.frames[self.pos]
        if not backsub:
            return True, self.frame, self.grayframe
        mask = threshold_mask(self.frame, self.threshold)
        back = mask * self.frame
        return True, back, back, mask

    def current_frame(self, backsub=True):
        return threshold_mask(self.frame, self.threshold) * self.frame if backsub else self.frame

    gray_frame = current_frame

    def backsub(self, im=None):
        if im is None:                                 # (mask, contours, signed distance function): DistMesh's input
            from .imgproc import findObjectThreshold
            return findObjectThreshold(self.frame, threshold=self.threshold)
        mask = threshold_mask(self.frame, self.threshold)
        im = np.asarray(im)
        return mask * im if im.ndim == 2 else mask[:, :, None] * im

    def isOpened(self):
        return self.pos < self.frames.shape[0]

    # frame source protocol of FlowEKFPipeline: random access to (raw frame, mask, frame shown to the filter)
    def __len__(self):
        return self.frames.shape[0]

    @property
    def shape(self):
        return self.frames.shape[1:3]

    def frame_at(self, f):
        fr = self.frames[f]
        mask = threshold_mask(fr, self.threshold)
        return fr, mask, mask * fr

    def release(self):
        self.pos = self.frames.shape[0]


class DeviceBuffer:
    """A block of device memory owned by the caller side of the C-ABI (hm_dev_alloc)."""

    def __init__(self, nbytes, device=0):
        self.device, self.nbytes = int(device), int(nbytes)
        p = _lib.c_vp()
        _lib.check(_lib.lib().hm_dev_alloc(self.device, self.nbytes, p), "hm_dev_alloc")
        self.ptr = p.value
        _lib.register(self, 4)

    def upload(self, a, offset=0):
        a = np.ascontiguousarray(a)
        assert offset + a.nbytes <= self.nbytes
        _lib.check(_lib.lib().hm_dev_upload(self.device, self.ptr + offset, _lib.ptr(a), a.nbytes), "hm_dev_upload")

    def download(self, out, offset=0):
        assert out.flags["C_CONTIGUOUS"] and offset + out.nbytes <= self.nbytes
        _lib.check(_lib.lib().hm_dev_download(self.device, _lib.ptr(out), self.ptr + offset, out.nbytes), "hm_dev_download")
        return out

    def close(self):
        if getattr(self, "ptr", None):
            _lib.lib().hm_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ArraySource:
    """Frame source over host arrays: video, masks (frames, H, W) uint8, optionally the frames the filter is
    shown when they differ from the ones the flow is computed on."""

    def __init__(self, video, masks, observed=None):
        video = np.ascontiguousarray(video, np.uint8)
        masks = np.ascontiguousarray(masks, np.uint8)
        if video.ndim != 3 or masks.shape != video.shape:
            raise ValueError("video and masks must both be (frames, H, W) uint8")
        if observed is not None:
            observed = np.ascontiguousarray(observed, np.uint8)
            if observed.shape != video.shape:
                raise ValueError("observed must have the shape of video")
        self.video, self.masks, self.observed = video, masks, observed

    def __len__(self):
        return self.video.shape[0]

    @property
    def shape(self):
        return self.video.shape[1:3]

    def frame_at(self, f):
        return self.video[f], self.masks[f], None if self.observed is None else self.observed[f]


class FrameRing:
    """Ring of frame slots in HBM fed from a frame source over a copy stream.

    Frame f lives in slot f % R of each plane (raw frame, mask, and -- when the source has one -- the frame
    shown to the filter).  A flow series reads nb + 1 consecutive frames through one base pointer, so the first
    `extra` slots are mirrored behind the ring (frames that land in slots 0 .. extra-1 are copied twice): any run of up to extra + 1
    frames is contiguous.  Host frames are staged in page-locked memory and copied with hipMemcpyAsync on a
    stream of their own; `sync()` makes what has been queued visible to the other streams (the caller waits
    before it launches work that reads them).  With slots >= len(source) nothing is ever overwritten (the
    resident mode)."""

    def __init__(self, source, slots, extra, device, with_observed):
        self.src, self.device = source, int(device)
        self.F = len(source)
        self.H, self.W = source.shape
        self.n = self.H * self.W
        self.R = int(min(slots, self.F))
        self.extra = int(extra) if self.R < self.F else 0
        total = self.R + self.extra
        self.d_video = DeviceBuffer(total * self.n, device)
        self.d_masks = DeviceBuffer(total * self.n, device)
        self.d_observed = DeviceBuffer(total * self.n, device) if with_observed else self.d_video
        self.planes = 3 if with_observed else 2
        self._lock = threading.RLock()
        self._stream = _lib.c_vp()
        _lib.check(_lib.lib().hm_copy_stream_create(self.device, ctypes.byref(self._stream)), "hm_copy_stream_create")
        self._nstage = max(2, self.extra + 2) if self.R < self.F else min(self.F, 16)                    # frames staged between two waits for the copy stream
        self._stage = _lib.c_vp()
        _lib.check(_lib.lib().hm_host_alloc(self._nstage * self.planes * self.n, ctypes.byref(self._stage)), "hm_host_alloc")
        self._stage_np = np.ctypeslib.as_array(ctypes.cast(self._stage, ctypes.POINTER(ctypes.c_uint8)),
                                               shape=(self._nstage, self.planes, self.n))
        self._staged = 0
        self.synced_hi = 0
        _lib.register(self, 4)
        self.lo = self.hi = 0                                    # frames [lo, hi) are in the ring (queued or there)
        self.bytes_uploaded = 0

    def reset(self, first):
        with self._lock:
            self.sync()
            self.lo = self.hi = self.synced_hi = int(first)

    def ptr(self, plane, f):
        """Device address of frame f in plane 0 (raw), 1 (mask) or 2 (shown to the filter)."""
        buf = (self.d_video, self.d_masks, self.d_observed)[plane]
        return buf.ptr + (f % self.R) * self.n

    def run_ptr(self, f):
        """Base address of the raw frames f, f + 1, ... (contiguous for extra + 1 frames)."""
        return self.d_video.ptr + (f % self.R) * self.n

    def ensure(self, hi, keep_from):
        """Queue the uploads of the frames up to hi - 1; frames before keep_from may be overwritten."""
        L = _lib.lib()
        with self._lock:
            hi = min(int(hi), self.F)
            if hi - keep_from > self.R:
                raise RuntimeError("frame ring too small: frames %d..%d asked for, %d slots" % (keep_from, hi - 1, self.R))
            while self.hi < hi:
                f = self.hi
                if self._staged == self._nstage:                  # the staging block is full: wait for its copies
                    _lib.check(L.hm_copy_stream_sync(self.device, self._stream), "hm_copy_stream_sync")
                    self._staged = 0
                st = self._stage_np[self._staged]
                fr, mk, ob = self.src.frame_at(f)
                parts = [(self.d_video, fr), (self.d_masks, mk)]
                if self.planes == 3:
                    parts.append((self.d_observed, fr if ob is None else ob))
                s = f % self.R
                for i, (buf, a) in enumerate(parts):
                    st[i][:] = np.asarray(a, np.uint8).reshape(-1)
                    src = self._stage.value + (self._staged * self.planes + i) * self.n
                    _lib.check(L.hm_dev_upload_async(self.device, buf.ptr + s * self.n, src, self.n, self._stream), "hm_dev_upload_async")
                    if s < self.extra:                            # the mirror behind the ring
                        _lib.check(L.hm_dev_upload_async(self.device, buf.ptr + (self.R + s) * self.n, src, self.n, self._stream),
                                   "hm_dev_upload_async")
                    self.bytes_uploaded += self.n
                self._staged += 1
                self.hi = f + 1
            self.lo = max(self.lo, self.hi - self.R)

    def sync(self):
        with self._lock:
            hi = self.hi
            _lib.check(_lib.lib().hm_copy_stream_sync(self.device, self._stream), "hm_copy_stream_sync")
            self._staged = 0
            self.synced_hi = hi              # frames below this are in device memory for certain

    def close(self):
        if getattr(self, "_stream", None) is not None and self._stream:
            _lib.lib().hm_copy_stream_destroy(self.device, self._stream)
            self._stream = None
        if getattr(self, "_stage", None) is not None and self._stage:
            self._stage_np = None
            _lib.lib().hm_host_free(self._stage)
            self._stage = None
        for b in {id(b): b for b in (self.d_video, self.d_masks, self.d_observed)}.values():
            b.close()


class _Done:
    """Marks a series in flight that has been waited for already (its thread is gone)."""

    @staticmethod
    def join():
        return None


class FlowEKFPipeline:
    """Brox flow of the coming frames overlapped with the filter on the current one.

        pipe = FlowEKFPipeline(kf, video, masks)        # kf: any kalman.*KalmanFilter on the same device
        for k in range(len(video) - 1):
            e = pipe.step(k)                            # frame k+1: flow of (k, k+1), then kf.compute

    ``video``: a frame source (``VideoStream``, ``ArraySource``: ``len()``, ``.shape``, ``.frame_at(f)`` ->
    raw frame, mask, frame shown to the filter or None) or, with ``masks``, (frames, H, W) uint8 host arrays
    (masks in {0,1}).  Frames are read from it one by one and uploaded on a copy stream into a ring of
    ``3 flow_batch + 3`` frame slots (``FrameRing``): the upload of the frames of the next flow series is
    queued right after the launches of the current one.  ``resident=True`` uploads the whole video at
    construction instead.  The flow parameters default to the reference's (src/optical_flow_ext.cpp:453-488).
    The numbers are those of ``bf.calc(video[k], video[k+1])`` followed by
    ``kf.compute(video[k+1], flow, masks[k+1])``: a pair's flow does not depend on the series it is computed in.
    """

    def __init__(self, kf, video, masks=None, flow_batch=8, device=0, brox_params=None, sor_threads=0, maskflow=True,
                 observed=None, return_flow=False, cu_reserve=32, concurrent_series=True, resident=False):
        """return_flow: whether step() brings the rendered flow planes of every frame to the host, as
        KalmanFilter.compute does for the reference's callers (8 MB per 1024^2 frame); a frame loop that
        looks at the error sums and the state only (reference run_kalmanfilter.py:78-89 ignores the return
        value altogether) leaves them on the device.
        observed: the frames the filter is shown, when they differ from the ones the flow is computed on
        (the reference CLI tracks the background-subtracted frame, renderer.py:770-773, while its flow tool
        works on the raw video); default: the video itself."""
        if hasattr(video, "frame_at"):
            if masks is not None or observed is not None:
                raise ValueError("a frame source brings its own masks / observed frames")
            source = video
            with_observed = source.frame_at(0)[2] is not None
        else:
            source = ArraySource(video, masks, observed)
            with_observed = observed is not None
        self.source = source
        self.kf, self.maskflow = kf, maskflow
        kf.return_flow = bool(return_flow)
        self.F = len(source)
        self.H, self.W = source.shape
        self.B = max(1, int(flow_batch))
        self.device = int(device)
        n = self.H * self.W
        self._px = n
        # the filter works on frame k + 1 while up to two series (of at most B pairs each, B + 1 frames) are in
        # flight or ready ahead of it and the next one's frames are being uploaded
        self.ring = FrameRing(source, self.F if resident else 3 * self.B + 3, self.B, device, with_observed)
        if resident:
            self.ring.ensure(self.F, 0)
            self.ring.sync()
        self.resident = bool(resident)
        _lib.register(self, 0)
        self.d_u = DeviceBuffer(3 * self.B * n * 4, device)          # flow planes: one buffer in use, two being filled
        self.d_v = DeviceBuffer(3 * self.B * n * 4, device)
        # concurrent_series: two flow handles (streams), two series in flight at a time -- while the filter works through
        # the pairs of one series the next TWO run beside it, each on a handle of its own.  A phase starts with one pair
        # (the whole chip) and a second series beside it, and the series grow as _next_concurrent sizes them (1, 2, 2, 3,
        # 5, 8 at 1024^2 / 201 vertices): the filter waits ~7 ms for its first pair instead of ~13 for a first series of
        # five, and once the series are full it does not wait at their boundaries any more (one series at a time: 0.3 to
        # 2.7 ms at every boundary, profiles/r04_frame_trace_64.txt).  Driver's bench, 20 frames: 267-270 against
        # 261-264 frames/s, 64 frames: 331-333 against 322-325.  (Round 2 measured this mode at 182 against 214 frames/s:
        # before the filter's streams had their priority and the flow's its CU mask.)  The results are the same bits
        # either way (tools/determinism_check.py).
        def make_handle():
            bf = _brox.BroxOpticalFlow(self.W, self.H, max_batch=self.B, device=device, **(brox_params or {}))
            bf.tune("sor_threads", sor_threads)         # 0: chosen per series (1024 for one or two pairs, else 512)
            if cu_reserve:
                # the flow streams leave some compute units alone: the filter's short dependent launches find room at
                # once while a series fills the rest (bench: 250 -> 254 frames/s with 32 of 256 reserved)
                bf.tune("cu_reserve", int(cu_reserve))
            return bf
        self._make_handle = make_handle
        self.cu_reserve = int(cu_reserve or 0)
        self.concurrent_series = bool(concurrent_series)
        self.bfs = [make_handle() for _ in range(2 if concurrent_series else 1)]
        self.bf = self.bfs[0]
        # split_start: a phase whose series times are known (an earlier phase measured them) starts with TWO series side by
        # side, on two handles -- a short one (two pairs) so that the filter, which has nothing to do until the first
        # flow arrives, starts soon, and the one _first_series() sizes behind it.  A lone short series leaves most of the
        # chip idle (it is a chain of ~500 short launches), so the second should cost the first little.  After that one
        # series at a time, as before.  The second handle is created when it is first needed.  Measured at 1024^2 / 201
        # vertices, 20 frames: 212 frames/s against 229 without (the two series slow each other and the filter's first
        # frames more than the earlier start gains: 0.92 instead of 0.73 ms per frame of waiting for flow); 64 frames: 286
        # against 292.  Again at the end of round 3 (calibrated series sizes, the second handle created before the timed region):
        # 249.5 against 254.4 frames/s at 20 frames, three runs each.  Off by default; the results are the same bits either way.
        self.split_start = False
        self.t_flow = self.t_ekf = 0.0
        self.iters = 0
        self.frame_done = []             # perf_counter() at the end of every step (steady-state rates)
        self.profile_full, self.profiled_pairs = False, 0    # set profile_full: the next series of flow_batch pairs is profiled
        self.profile_min_pairs = None    # ... or of at least this many (a series that runs beside the filter, not a phase's first)
        self.trace = None                # callable(str) for per-frame scheduling messages
        # pairs [lo, hi) of `ready` have their flow in buffer `buf`; the series in `_flying` (oldest first; each a dict
        # lo, hi, buf, handle, thread) are being computed while the filter works on `ready`
        self._ready, self._buf, self._flying, self._thread_exc = (0, 0), 0, [], None
        self._end = self.F - 1
        self._cursor = 0                 # the oldest frame still needed (the pair the filter is at)
        self._series_s = {}              # pairs -> seconds a series of that size took alone (first series of phases)
        self._frame_s = None             # seconds per frame of the filter (median of the last frames)
        self._frame_hist = []
        self._calibrated = False
        self._flow_late = False
        # the series beside a phase's opening pair starts this fraction of (a series of two pairs alone) later: 0.3 x 7 ms.
        # Three runs each, 20 frames: 263-264 frames/s with 0 (opening wait 7.0-7.7 ms), 263-274 with 0.2, 266-268 with 0.35
        # (6.1-6.4 ms), 262-266 with 0.5 (the second series is late instead)
        self.second_delay = 0.3
        self.adaptive_first = True       # size the first series of a phase from those measurements
        self.first_series = 0            # > 0: fixed size of the first series of every phase
        self.profiled_handle = None
        self.gc_freeze = True            # see run()
        # model_ramp: the first series of a phase and the ones after it are sized from measured series / frame times
        # (calibrate: what a series of 2 and of B pairs takes alone; _first_series: the smallest first series the ramp can
        # follow; _next_series: the largest series that is done when the filter is through with the previous one, at least
        # one pair more than that one) instead of a fixed 1.7 x ramp from a first series sized for a full one behind it.
        # Driver's 20-frame bench, three runs each on one box: 255.0 against 244.0 frames/s (waiting for flow 0.62
        # against 0.92 ms per frame: a first series of 4 or 5 pairs instead of 8), 64 frames: 307.7 against 301.3.
        self.model_ramp = True

    # -- flow series ---------------------------------------------------------------------------------
    def _launch(self, k, end, most, alone=None, whole=None, delay=0.0):
        """Queue the series of pairs k .. k + nb - 1 on a handle and a buffer that are free (`delay`: seconds the helper
        thread waits before it queues the launches)."""
        nb = min(most, end - k)
        n, B = self._px, self.B
        busy_h = {f["handle"] for f in self._flying}
        busy_b = {f["buf"] for f in self._flying} | {self._buf}
        h = next(i for i in range(len(self.bfs)) if i not in busy_h)
        buf = next(i for i in range(3) if i not in busy_b)
        bf = self.bfs[h]

        def work():
            try:
                if delay > 0:
                    time.sleep(delay)
                if self.profile_full and self.profiled_pairs == 0 and nb >= (self.profile_min_pairs or self.B) and not first:
                    bf.profile(True)                    # hm_brox_profile around the first series of flow_batch pairs
                    self.profiled_pairs, self.profiled_handle, self.profile_full = nb, bf, False
                elif bf is self.profiled_handle:
                    bf.profile(False)                   # totals stay readable (hm_brox_profile_read)
                # frames k .. k + nb: normally queued for upload when the previous series was launched; the copy
                # stream is waited for here, on the helper thread, before the flow stream is given work that reads them
                self.ring.ensure(k + nb + 1, self._cursor)
                self.ring.sync()
                if self.cu_reserve:
                    # a series the filter waits for with nothing to do (the first of a phase) gets the whole chip, the
                    # others leave `cu_reserve` compute units to the filter
                    bf.tune("whole_chip", 1 if (first if whole is None else whole) else 0)
                bf.calc_dev(nb, self.ring.run_ptr(k), self.ring.run_ptr(k) + n,
                            self.d_u.ptr + buf * B * n * 4, self.d_v.ptr + buf * B * n * 4)
                # ... and while this series runs, the frames of the next one go up (reference run_kalmanfilter.py:78-89
                # reads one frame per iteration; here the read-ahead is one series)
                nxt = min(self._end, k + nb + self._next_series(nb))
                self.ring.ensure(min(nxt + 1, self._cursor + self.ring.R), self._cursor)
            except Exception as exc:                    # noqa: BLE001 -- re-raised by the thread that waits
                self._thread_exc = exc
        first = not self._flying and self._ready[0] == self._ready[1]       # the first series of a phase: nothing runs beside it
        if alone is not None:
            first = alone
        measured = first and not self.concurrent_series      # (with two handles the second series starts beside the first)
        t = threading.Thread(target=work)
        t0 = time.perf_counter()
        t.start()
        self._flying.append({"lo": k, "hi": k + nb, "buf": buf, "handle": h, "thread": t, "t0": t0, "alone": measured})

    def _next_series(self, last):
        """Pairs of the series that follows one of `last` pairs.  A phase starts with two pairs and grows 2, 3, 5, 8: a
        series of n 1024^2 pairs takes about 5.3 + 1.45 (n - 1) ms, a frame of the filter about 4 ms, and this ramp
        is the one with the least waiting for that pair of numbers (1, 2, 4, 8 waits 40 % longer); a ramp that is too
        steep only costs the wait for its larger series, once.  (With two handles: 1, 2, 3, 5, 8.)"""
        guess = min(self.B, max(last + 1, int(1.7 * last)))
        model = self._series_model()
        if model is None or not self.adaptive_first or not self.model_ramp:
            return guess
        # with measurements: the largest series that is done (beside the filter: ~1.3 x as long as alone) by the time the
        # filter is through with the `last` frames it has just been given -- a steeper ramp has the filter wait for it --
        # but at least one pair more than the last one: larger series cost less per pair, and the wait for a series that
        # is one pair too large is a millisecond or two, once
        a, b, n_a = model
        room = last * self._frame_s / 1.3
        n = int(n_a + (room - a) / b) if b > 0 else self.B
        return min(self.B, max(last + 1, min(n, 2 * last)))

    def _next_concurrent(self):
        """Pairs of the next series with two handles (concurrent_series): the largest series that is done by the time the
        filter needs its first pair.  That moment follows from what is queued in front of it -- the pairs that are ready
        and not yet used, and the series in flight, each ready when the work of the series in front of it and its own are
        done (they share the chip; 1.6 x what the work takes alone: beside the filter) and then used up at one frame of
        the filter per pair.
        None without measurements (calibrate / earlier frames): the fixed ramp of _next_series."""
        # (a ramp sized this way is 1, 2, 2, 3, 3, 3, 4 at 1024^2 / 201 vertices: 9.5 ms of waiting over the driver's 20 frames,
        # 6.4 of them for the opening pair, instead of the 13 + 2 of one series at a time)
        model = self._series_model()
        if model is None or not self.adaptive_first or not self.model_ramp:
            return None
        a, b, n_a = model
        F, slow = max(self._frame_s, 1e-4), 1.6
        now = time.perf_counter()
        t = now + max(0, self._ready[1] - self._cursor) * F
        last, overdue = max(1, self._ready[1] - self._ready[0]), bool(self._flow_late)
        ahead = 0.0                                # what the series in flight still have to do, in time alone
        for f in self._flying:
            n = f["hi"] - f["lo"]
            alone = a + b * (n - n_a)
            overdue = overdue or f["t0"] + slow * alone < now
            # (the series in flight share the chip: one is done when the work in front of it and its own are)
            ahead += max(0.0, alone - (now - f["t0"]) / slow)
            t = max(t, now + slow * ahead) + n * F
            last = n
        best = 1
        for n in range(1, self.B + 1):
            if now + slow * (ahead + a + b * (n - n_a)) <= t:
                best = n
        fit = best
        best = max(best, last, min(self.B, 2))      # (a series of one pair costs 5.3 ms a pair, one of two 3.4: only the opening one)
        if overdue:
            best = max(best, last + 1, int(1.7 * last))
        if self.trace:
            self.trace("next series: %.1f ms until it is needed, a series of n takes %.2f + %.2f (n - %d) ms alone, a frame %.2f ms: "
                       "%d pairs fit, last %d%s -> %d" % (1e3 * (t - now), 1e3 * a, 1e3 * b, n_a, 1e3 * F, fit, last,
                                                          ", overdue" if overdue else "", min(self.B, best)))
        return min(self.B, best)

    def _series_model(self):
        """(a, b, n_a): a series of n pairs alone takes about a + b (n - n_a) seconds, from the first series of earlier
        phases (see _first_series); None without measurements."""
        ts = self._series_s
        if not ts or self._frame_s is None:
            return None
        (n_a, t_a), (n_b, t_b) = sorted(ts.items())[0], sorted(ts.items())[-1]
        if n_b > n_a:
            b = max(0.0, (t_b - t_a) / (n_b - n_a))
        else:
            b = 0.2 * t_a / (1.0 + 0.2 * (n_a - 2))
        return t_a, b, n_a

    def _wait(self, f):
        f["thread"].join()
        if self._thread_exc is not None:
            exc, self._thread_exc = self._thread_exc, None
            raise exc
        t1 = time.perf_counter()
        self.bfs[f["handle"]].sync()
        t2 = time.perf_counter()
        if f.get("alone") and t2 - t1 > 2e-4:
            # the caller really waited for this series and nothing ran beside it (the first series of a phase): launch ->
            # completion is what a series of that many pairs takes -- kept for sizing the first series of later phases
            n = f["hi"] - f["lo"]
            self._series_s[n] = min(self._series_s.get(n, 1e9), t2 - f["t0"])

    def _first_series(self):
        """Pairs of the first series of a phase.  The filter waits for that series with nothing to do, so it should be
        just large enough that the series after it (of B pairs, running beside the filter) is done by the time the
        filter is through with these: n1 x (time of a frame of the filter) >= time of a series of B pairs.  Both times
        are measured in earlier phases (a series of n pairs takes about a + b (n - 1): two first series of different
        sizes give a and b, one gives a with b taken as a fifth of a two-pair series -- 1.45 of 7.2 ms at 1024^2; a
        series beside the filter takes ~1.3 x as long as alone); without measurements a phase starts with two pairs."""
        if self.concurrent_series:
            return 1
        default = min(2, self.B)
        if self.first_series:
            return max(1, min(self.B, int(self.first_series)))
        model = self._series_model()
        if model is None or not self.adaptive_first:
            return default
        t_a, b, n_a = model
        F = max(self._frame_s, 1e-4)
        if self.model_ramp:
            # the smallest first series the ramp of _next_series can follow without shrinking: a series of as many pairs,
            # running beside the filter (1.3 x), is done when the filter is through with these
            # (1.25, not the 1.3 of the series that follow: at 1024^2 / 201 vertices the choice between 4 and 5 pairs sat
            # exactly on the measured frame time -- 3.17 ms -- and fell either way from run to run: 258-260 frames/s with
            # 4, 250-252 with 5 over the driver's 20 frames)
            for n1 in range(default, self.B + 1):
                if 1.25 * (t_a + b * (n1 - n_a)) <= n1 * F:
                    return n1
            return self.B
        # the series after the first need not be a full one (_next_series ramps up to B): sized for one of at most 8 pairs
        t_full = t_a + b * (min(self.B, 8) - n_a)
        n1 = int(np.ceil(1.3 * t_full / F))
        return max(default, min(self.B, n1))

    def flow_sync(self):
        """Wait for every series in flight (their results stay where they are)."""
        for f in self._flying:
            self._wait(f)
            f["thread"] = _Done

    def calibrate(self, first=0):
        """What a flow series of 2 and of B pairs takes with nothing beside it (seconds, kept in _series_s for
        _first_series / _next_series): three calls on the frames first .. first + B, the first of which also absorbs the
        one-time costs of a handle's first launches (measured as part of a phase's first series they made a series of 2
        pairs look like 9.7 instead of 6.8 ms, and every series after it was sized from that).  ~30 ms at 1024^2, once
        per pipeline, at the start of its first phase."""
        nb = min(self.B, self.F - 1 - first)
        if nb < 2:
            return
        n, B = self._px, self.B
        self.ring.ensure(first + nb + 1, first)
        self.ring.sync()
        bf = self.bfs[0]
        if self.cu_reserve:
            bf.tune("whole_chip", 1)
        for pairs, keep in ((2, False), (2, True), (nb, True)):
            t0 = time.perf_counter()
            bf.calc_dev(pairs, self.ring.run_ptr(first), self.ring.run_ptr(first) + n, self.d_u.ptr, self.d_v.ptr)
            bf.sync()
            if keep:
                self._series_s[pairs] = min(self._series_s.get(pairs, 1e9), time.perf_counter() - t0)
        self._calibrated = True

    def begin(self, first=0, end=None):
        """Start a phase: the pairs first .. end-1 will be asked for in order."""
        self.flow_sync()
        self._flying = []
        self._end = self.F - 1 if end is None else min(int(end), self.F - 1)
        self._ready = (first, first)
        self._cursor = first
        self._flow_late = False
        if not self.resident:
            self.ring.reset(first)               # nothing of an earlier phase is assumed to be in the ring
        if self.model_ramp and self.adaptive_first and not self._calibrated:
            self.calibrate(first)                # (the frames it uploads are the first ones of this phase: they stay)

    def _top_up(self):
        """Keep as many series in flight as there are handles (one unless concurrent_series) -- two at the very start of
        a phase when split_start applies."""
        starting = not self._flying and self._ready[0] == self._ready[1]
        if (starting and self.split_start and not self.concurrent_series and self._series_s and self._frame_s is not None
                and self.B >= 3 and self._end - self._ready[1] >= 4 and not self.first_series):
            if len(self.bfs) < 2:
                self.bfs.append(self._make_handle())
            k = self._ready[1]
            n1 = max(3, self._first_series())
            self._launch(k, self._end, 2, alone=False, whole=True)
            self._launch(k + 2, self._end, n1, alone=False, whole=True)
            return
        limit = len(self.bfs) if self.concurrent_series else 1
        while len(self._flying) < limit:
            last = self._flying[-1] if self._flying else None
            nxt = last["hi"] if last else self._ready[1]
            if nxt >= self._end:
                return
            size = (last["hi"] - last["lo"]) if last else (self._ready[1] - self._ready[0])
            most = self._next_series(size) if size else self._first_series()
            delay = 0.0
            if self.concurrent_series and size:
                most = self._next_concurrent() or most
                if last is not None and self._ready[0] == self._ready[1] and last.get("opening") and self.second_delay:
                    # the series beside the opening one: the opening pair -- what the filter is waiting for, a chain of
                    # ~450 short launches -- has the chip to itself for a while first (the second series is needed a frame
                    # of the filter after that pair, and the two slow each other down)
                    model = self._series_model()
                    if model is not None:
                        delay = float(self.second_delay) * model[0]
            self._launch(nxt, self._end, most, delay=delay)
            if not last and self._ready[0] == self._ready[1]:
                self._flying[-1]["opening"] = True

    def flow_ready(self, k):
        """Make the flow of pair (k, k+1) available -> (device pointer of u, of v)."""
        if not (0 <= k < self.F - 1):
            raise IndexError("pair %d of a %d-frame video" % (k, self.F))
        if not (self._ready[0] <= k < self._ready[1]):
            if k != self._ready[1] or (self._flying and self._flying[0]["lo"] != k):
                self.begin(k, max(self._end, k + 1))     # random access: start over from k
            elif k >= self._end:
                self._end = min(self.F - 1, k + 1)       # past the end of the phase run() announced: go on pair by pair
            self._cursor = k
            self._top_up()
            f = self._flying.pop(0)
            t_w = time.perf_counter()
            opening = self._ready[0] == self._ready[1]             # the first series of a phase is always waited for
            if f["thread"] is not _Done:
                self._wait(f)
            # the filter waited for this series although it had others to work through before (_next_concurrent)
            self._flow_late = (not opening) and time.perf_counter() - t_w > 5e-4
            self._ready, self._buf = (f["lo"], f["hi"]), f["buf"]
            self._top_up()
        self._cursor = k
        i = k - self._ready[0]
        off = (self._buf * self.B + i) * self._px * 4
        return self.d_u.ptr + off, self.d_v.ptr + off

    def flow_host(self, k):
        """The flow of pair (k, k+1) as an (H, W, 2) host array (for writing .mat files, tests)."""
        pu, pv = self.flow_ready(k)
        out = np.empty((2, self.H, self.W), np.float32)
        _lib.check(_lib.lib().hm_dev_download(self.device, _lib.ptr(out[0]), pu, self._px * 4), "hm_dev_download")
        _lib.check(_lib.lib().hm_dev_download(self.device, _lib.ptr(out[1]), pv, self._px * 4), "hm_dev_download")
        return np.dstack((out[0], out[1]))

    # -- the frame loop of reference run_kalmanfilter.py:78-89 ----------------------------------------------
    def step(self, k):
        """Frame k+1: flow of (k, k+1) -- usually already there -- then kf.compute on it."""
        import time
        if not (0 <= k < self.F - 1):
            raise IndexError("pair %d of a %d-frame video" % (k, self.F))
        t0 = time.perf_counter()
        pu, pv = self.flow_ready(k)
        t1 = time.perf_counter()
        n = self._px
        # the next frame's mask, when its upload has been waited for already (every flow series is launched behind a wait
        # for the uploads of its frames): the filter queues that mask's outline a frame ahead
        nxt = self.ring.ptr(1, k + 2) if (k + 2 < self.F and self.ring.lo <= k + 2 < self.ring.synced_hi) else None
        obs = DeviceObservation(self.ring.ptr(2, k + 1), pu, pv, self.ring.ptr(1, k + 1),
                                y_m_host=self.source.frame_at(k + 1)[1], next_mask=nxt)
        e = self.kf.compute(obs, None, None, maskflow=self.maskflow)
        t2 = time.perf_counter()
        # what a frame of the filter takes: the median of the last few (the first frames of a filter's life carry
        # first-use costs -- allocations, the first launches -- that a running mean drags along for a whole phase)
        self._frame_hist.append(t2 - t1)
        if len(self._frame_hist) > 8:
            self._frame_hist.pop(0)
        # (with fewer than five on record the fastest one: the slow ones are the first-use frames)
        hist = sorted(self._frame_hist)
        self._frame_s = (0.5 * (hist[(len(hist) - 1) // 2] + hist[len(hist) // 2])) if len(hist) >= 5 else hist[0]
        self.t_flow += t1 - t0
        self.t_ekf += t2 - t1
        self.iters += getattr(self.kf, "niter", 1)
        self.frame_done.append(t2)
        if self.trace:
            self.trace("step %d: flow wait %.2f ms, filter %.2f ms (%d iterations), series ready %s in flight %s"
                       % (k, 1e3 * (t1 - t0), 1e3 * (t2 - t1), getattr(self.kf, "niter", 1), self._ready,
                          [(f["lo"], f["hi"]) for f in self._flying]))
        return e

    def run(self, first=0, end=None, on_frame=None):
        """compute() for the frames first+1 .. end; on_frame(k, error_tuple) after each.

        gc_freeze (attribute, default True; the name is round 3's): the interpreter's automatic collections are switched
        off for the phase (gc.disable) and switched back on at its end if they were on -- a full collection, which the
        frame loop's allocations trigger every ~17 frames, took 6.7 ms inside the Python wrapper of hm_update_run with
        numpy / scipy loaded (one frame in 17 at 8.4 instead of 1.7 ms; longer with torch); the youngest generation is
        collected by hand every 64 frames, which takes microseconds.  Nothing the caller froze or tuned is touched
        (round 3 called gc.freeze() / gc.unfreeze() here, which unfroze the caller's objects as well)."""
        end = self.F - 1 if end is None else min(int(end), self.F - 1)
        import gc
        quiet = bool(self.gc_freeze) and gc.isenabled()
        if quiet:
            gc.disable()
        try:
            self.begin(first, end)
            for k in range(first, end):
                e = self.step(k)
                if on_frame is not None:
                    on_frame(k, e)
                if quiet and (k - first) % 64 == 63:
                    gc.collect(0)
        finally:
            if quiet:
                gc.enable()

    def close(self):
        """Joins the helper threads of the series in flight, drains and destroys the copy stream, frees the ring, the
        flow planes and the flow handles (idempotent).  The filter is the caller's."""
        if getattr(self, "_closed", False):
            return
        self._closed = True
        try:
            self.flow_sync()
        finally:
            self.ring.close()
            for b in (self.d_u, self.d_v):
                b.close()
            for bf in self.bfs:
                bf.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
