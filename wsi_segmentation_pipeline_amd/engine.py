"""Host side of the HIP inference path: weight prepack, workspaces and launch wrappers.

PyTorch is used only as plumbing (device memory, streams); all compute goes through
libwsi_hip.so (include/wsi_hip.h).  There is no CPU fallback: tensors must live on a GPU.
"""
import ctypes as C
import weakref

import numpy as np
import torch

from . import native

BN_EPS = 1e-5                                   # nn.BatchNorm2d default (reference resnets_shift.py:117)
PARITY, SPEED, MX = 2, 1, 3                     # planes: fp16 hi + fp16 lo pair (3 passes; r01-r04: bf16 pair) / single bf16 / fp16 + MX-fp6 cross terms
AUTO = 'auto'                                   # AutoTrunkEngine: mx unless a stratified two-mode probe of the slide says parity


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(st=None):
    return C.c_void_p((st or torch.cuda.current_stream()).cuda_stream)


def _f32(sd, key):
    return np.ascontiguousarray(sd[key].detach().to('cpu', torch.float32).numpy())


def normalize_lut(mean, std):
    """3x256 fp32 table of the eval transform (reference utils/preprocessing.py:209-212)."""
    lib = native.load()
    m = np.asarray(mean, np.float32)
    s = np.asarray(std, np.float32)
    out = np.empty((3, 256), np.float32)
    native.check(lib.wsi_normalize_u8_lut(_np_ptr(m), _np_ptr(s), _np_ptr(out)), 'wsi_normalize_u8_lut')
    return out


def _require_gpu(t, what):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError('%s must be a GPU tensor: the WSI inference path runs on HIP kernels only '
                           '(no CPU fallback)' % what)


def batch_sizes(n, cap, h=256, w=256):
    """Batch sizes (each <= cap) for n images of h x w.  Several batches are sized in whole ROUNDS of the chip where that fits the
    cap: a batch of q = 512 * 256^2 / (h w) images gives every trunk launch a whole number of workgroup rounds (layer 4: one
    workgroup per 256^2-pixel image and channel block, 512 resident), so all batches but the last are multiples of q and the last
    takes the rest (24 648 tiles, cap 6 656: 3 x 6 144 + 6 216 - a 12.04-round launch runs 13 rounds; measured r03: +0.2 % on cfg3, inside the
    run-to-run noise - the nearly empty thirteenth round is short).
    Otherwise equal sizes without a short tail, as before."""
    n, cap = int(n), max(1, int(cap))
    k = max(1, -(-n // cap))
    if k == 1:
        return [n] if n else []
    q = max(1, 512 * 65536 // max(h * w, 1))
    base = int(round(n / k / q)) * q
    if base >= q and n - base * (k - 1) <= cap and n - base * (k - 1) > 0 and base <= cap:
        return [base] * (k - 1) + [n - base * (k - 1)]
    mb = -(-n // k)                                        # equal-sized batches: ceil(n / ceil(n / cap))
    return [min(mb, n - i) for i in range(0, n, mb)]


class TrunkEngine:
    """ResNet-18 trunk (stem + layer1..4) of the reference ``resnets_shift.ResNet`` on HIP kernels,
    with an optional Linear(512->K) head fused after the average pool (``fc0`` or ``Classifier``).

    state_dict: reference key names (conv1.weight, bn1.*, layerL.B.convK.weight, ...).
    """

    def __init__(self, state_dict, device, planes=MX, head=None, max_batch=None,
                 mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), streams=2):
        self.lib = native.load()
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('TrunkEngine needs a GPU device, got %s' % device)
        if planes not in (1, 2, 3):
            raise ValueError('planes must be 1 (speed), 2 (parity, fp16 pair) or 3 (mx, fp16 + MX-fp6)')
        self.planes = planes
        # images per trunk call; None = the tuned size (6200 patches of 256 x 256, scaled by patch area: what bench.py measures) as far
        # as the free HBM allows - the workspace is ~8.3 MB per 256 x 256 patch, one per stream slot (`_auto_cap`; r01-r04: 2000 and
        # one stream, i.e. a caller of the drop-in modules did not get the benchmarked configuration)
        self.max_batch = None if max_batch is None else int(max_batch)
        # batches of one call are spread round-robin over `streams` HIP streams (own workspace each), so a
        # memory-bound stage of one batch overlaps an MFMA-bound stage of another and grid tails get filled
        self._streams = [torch.cuda.Stream(device=self.device) for _ in range(max(1, int(streams)))] if streams > 1 else []
        self._keep = []                          # device tensors referenced by raw pointers
        self._ws = {}
        self.wt = native.WsiTrunkWeights()
        self.wt.planes = planes
        sd = state_dict

        def bn(prefix):
            return [_f32(sd, prefix + s) for s in ('.weight', '.bias', '.running_mean', '.running_var')]

        def dev(a):
            t = torch.from_numpy(a).to(self.device)
            self._keep.append(t)
            return t

        # stem
        w = _f32(sd, 'conv1.weight')
        pk = np.empty(self.lib.wsi_prepack_stem_bytes(planes), np.uint8)
        bias = np.empty(64, np.float32)
        g, b, m, v = bn('bn1')
        native.check(self.lib.wsi_prepack_stem(_np_ptr(w), _np_ptr(g), _np_ptr(b), _np_ptr(m), _np_ptr(v), BN_EPS,
                                               planes, _np_ptr(pk), _np_ptr(bias)), 'wsi_prepack_stem')
        self.wt.stem_w = dev(pk).data_ptr()
        self.wt.stem_b = dev(bias).data_ptr()
        if planes >= 2:                          # u8 slide input: transform folded into the weights (exact integer pixels)
            pk8 = np.empty(self.lib.wsi_prepack_stem_bytes(2), np.uint8)
            bias8 = np.empty(64, np.float32)
            mean_a, std_a = np.asarray(mean, np.float32), np.asarray(std, np.float32)
            native.check(self.lib.wsi_prepack_stem_u8(_np_ptr(w), _np_ptr(g), _np_ptr(b), _np_ptr(m), _np_ptr(v), BN_EPS,
                                                      _np_ptr(mean_a), _np_ptr(std_a), planes, _np_ptr(pk8), _np_ptr(bias8)),
                         'wsi_prepack_stem_u8')
            self.wt.stem_w_u8 = dev(pk8).data_ptr()
            self.wt.stem_b_u8 = dev(bias8).data_ptr()
            for i in range(3):
                self.wt.norm[i], self.wt.norm[3 + i] = float(mean_a[i]), float(std_a[i])

        def conv(wkey, bnkey, k):
            w = _f32(sd, wkey)
            cout, cin = w.shape[0], w.shape[1]
            pk = np.empty(self.lib.wsi_prepack_conv_bytes(cout, cin, k, planes), np.uint8)
            bias = np.empty(cout, np.float32)
            g, b, m, v = bn(bnkey)
            native.check(self.lib.wsi_prepack_conv(_np_ptr(w), _np_ptr(g), _np_ptr(b), _np_ptr(m), _np_ptr(v), BN_EPS,
                                                   cout, cin, k, planes, _np_ptr(pk), _np_ptr(bias)), 'wsi_prepack_conv')
            return dev(pk).data_ptr(), dev(bias).data_ptr()

        for L in range(1, 5):
            for B in range(2):
                for K in (1, 2):
                    p = 'layer%d.%d' % (L, B)
                    i = (L - 1) * 4 + B * 2 + (K - 1)
                    self.wt.conv_w[i], self.wt.conv_b[i] = conv('%s.conv%d.weight' % (p, K), '%s.bn%d' % (p, K), 3)
            if L > 1:
                p = 'layer%d.0.downsample' % L
                self.wt.down_w[L - 2], self.wt.down_b[L - 2] = conv(p + '.0.weight', p + '.1', 1)
        self.set_head(head)
        self.lut = dev(normalize_lut(mean, std))

    # ------------------------------------------------------------------ configuration
    def set_head(self, head):
        """head = (weight (K,512), bias (K,)) tensors/arrays or None."""
        if head is None:
            self.wt.head_w, self.wt.head_b, self.wt.head_k = None, None, 0
            self.head_k = 0
            return
        w = torch.as_tensor(head[0]).detach().to(self.device, torch.float32).contiguous()
        b = torch.as_tensor(head[1]).detach().to(self.device, torch.float32).contiguous()
        if w.dim() != 2 or w.shape[1] != 512 or b.shape[0] != w.shape[0]:
            raise ValueError('head must be Linear(512 -> K)')
        self._head = (w, b)
        self.wt.head_w, self.wt.head_b, self.wt.head_k = w.data_ptr(), b.data_ptr(), int(w.shape[0])
        self.head_k = int(w.shape[0])

    def _workspace(self, n, h, w, slot=0):
        """(workspace, capacity): one workspace per (patch shape, stream slot), planned for the largest batch seen so far; it
        serves every smaller batch as is (wsi_trunk_forward's workspace_n), so ragged last batches and variable bag
        counts neither allocate nor re-zero ~8 MB per patch."""
        key = (h, w, slot)
        ent = self._ws.get(key)
        if ent is None or ent[1] < n:
            nbytes = self.lib.wsi_trunk_workspace_bytes(n, h, w, self.planes)
            if nbytes == 0:
                raise ValueError('unsupported patch shape %dx%d (need multiples of 32) or batch %d' % (h, w, n))
            self._drop_workspace(key)                       # a smaller plan of the same shape is released first
            if len(self._ws) >= 4 * max(1, len(self._streams)):   # keep the plan cache small
                self._drop_workspace(next(iter(self._ws)))
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
            native.check(self.lib.wsi_trunk_workspace_init(_ptr(ws), n, h, w, self.planes, _stream()),
                         'wsi_trunk_workspace_init')
            ent = self._ws[key] = (ws, n)
        return ent

    def _drop_workspace(self, key):
        ent = self._ws.pop(key, None)
        if ent is not None:                                 # the library forgets the address before the allocator can reuse it
            self.lib.wsi_trunk_workspace_release(_ptr(ent[0]))

    def release_workspaces(self):
        """Free every planned workspace (they are re-planned on the next forward)."""
        for key in list(self._ws):
            self._drop_workspace(key)

    def __del__(self):
        try:
            for key in list(getattr(self, '_ws', {})):
                self._drop_workspace(key)
        except Exception:                                   # interpreter shutdown: the library may be gone
            pass

    # ------------------------------------------------------------------ forward passes
    def _run(self, n, h, w, in_f32, slide, tile_xy, want_feat, want_logits, want_fmap, tap=None, slot=0):
        ws, cap = self._workspace(n, h, w, slot)
        dev = self.device
        feat = torch.empty((n, 512), dtype=torch.float32, device=dev) if want_feat else None
        logits = torch.empty((n, self.head_k), dtype=torch.float32, device=dev) if want_logits else None
        fmap = torch.empty((n, 512, h // 32, w // 32), dtype=torch.float32, device=dev) if want_fmap else None
        if slide is not None:
            sp, pitch, sh, sw = _ptr(slide), slide.stride(0), slide.shape[0], slide.shape[1]
        else:
            sp, pitch, sh, sw = None, 0, 0, 0
        if tap is not None:
            stage = (0, 1, 1, 2, 2, 3, 3, 4, 4)[tap]
            c = 64 << max(stage - 1, 0)
            hh = h >> (2 + max(stage - 1, 0))
            ww = w >> (2 + max(stage - 1, 0))
            out = torch.empty((n, c, hh, ww), dtype=torch.float32, device=dev)
            native.check(self.lib.wsi_trunk_forward_tap(C.byref(self.wt), _ptr(in_f32), sp, pitch, sh, sw, _ptr(tile_xy),
                                                        _ptr(self.lut), n, h, w, _ptr(ws), cap, tap, _ptr(out), _stream()),
                         'wsi_trunk_forward_tap')
            return out
        native.check(self.lib.wsi_trunk_forward(C.byref(self.wt), _ptr(in_f32), sp, pitch, sh, sw, _ptr(tile_xy),
                                                _ptr(self.lut), n, h, w, _ptr(ws), cap, _ptr(feat), _ptr(logits), _ptr(fmap),
                                                _stream()), 'wsi_trunk_forward')
        return feat, logits, fmap

    def forward_f32(self, x, feat=False, logits=False, fmap=False, tap=None):
        """x: (N,3,H,W) normalised fp32 on the GPU.  Returns (feat, logits, fmap) (None where not asked)."""
        _require_gpu(x, 'input batch')
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError('expected (N,3,H,W), got %s' % (tuple(x.shape),))
        if logits and not self.head_k:
            raise RuntimeError('no head set')
        x = x.to(torch.float32).contiguous()
        n, _, h, w = x.shape
        if tap is not None:
            return self._run(n, h, w, x, None, None, False, False, False, tap)
        return self._batched(n, lambda i, m, slot: self._run(m, h, w, x[i:i + m], None, None, feat, logits, fmap, slot=slot), h, w)

    def forward_tiles(self, slide_u8, tile_xy, ph, pw, feat=False, logits=True, fmap=False, tap=None):
        """slide_u8: (SH,SW,3) uint8 GPU tensor (last two dims contiguous); tile_xy: (N,2) int32 GPU tensor of
        top-left corners in slide pixels.  Tile read + transform are fused into the stem kernel."""
        _require_gpu(slide_u8, 'slide')
        _require_gpu(tile_xy, 'tile list')
        if slide_u8.dtype != torch.uint8 or slide_u8.dim() != 3 or slide_u8.shape[2] != 3 or slide_u8.stride(2) != 1 \
                or slide_u8.stride(1) != 3:
            raise ValueError('slide must be (H,W,3) uint8 with packed RGB pixels')
        if logits and not self.head_k:
            raise RuntimeError('no head set')
        tile_xy = tile_xy.to(torch.int32).contiguous()
        n = tile_xy.shape[0]
        if tap is not None:
            return self._run(n, ph, pw, None, slide_u8, tile_xy, False, False, False, tap)
        return self._batched(n, lambda i, m, slot: self._run(m, ph, pw, None, slide_u8, tile_xy[i:i + m], feat, logits, fmap,
                                                             slot=slot), ph, pw)

    TUNED_BATCH_256 = 6200                                   # images of 256 x 256 per trunk call (bench.py --batch default)

    def _auto_cap(self, h, w):
        """Default images per trunk call: the tuned batch scaled by patch area, limited so that the workspaces of all stream slots
        stay inside 60 % of the memory that is free now (plus what this engine already holds)."""
        want = max(1, int(self.TUNED_BATCH_256 * 65536 // max(h * w, 1)))
        per = self.lib.wsi_trunk_workspace_bytes(64, h, w, self.planes) / 64.0
        if per <= 0:
            return want
        try:                                                 # free = what the driver reports + what torch's caching allocator holds unused
            free = torch.cuda.mem_get_info(self.device)[0] + max(0, torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device))
        except Exception:
            return want
        held = sum(int(ws.numel()) for ws, _ in self._ws.values())
        slots = max(1, len(self._streams))
        return max(1, min(want, int(0.6 * (free + held) / (slots * per))))

    def _batched(self, n, run, h=256, w=256):
        """Split n images into max_batch chunks; with several chunks, alternate them over the side streams."""
        cap = self.max_batch if self.max_batch else self._auto_cap(h, w)
        sizes = batch_sizes(n, cap, h, w)
        starts = [sum(sizes[:j]) for j in range(len(sizes))]
        if len(sizes) > 1:                                  # plan every slot's workspace for the largest batch once (the last one)
            for slot in range(max(1, len(self._streams)) if self._streams else 1):
                self._workspace(max(sizes), h, w, slot)
        if len(starts) == 1 or not self._streams:
            outs = [run(i, m, 0) for i, m in zip(starts, sizes)]
        else:
            cur = torch.cuda.current_stream()
            ready = torch.cuda.Event()
            ready.record(cur)
            outs = []
            for j, i in enumerate(starts):
                st = self._streams[j % len(self._streams)]
                st.wait_event(ready)
                with torch.cuda.stream(st):
                    outs.append(run(i, sizes[j], j % len(self._streams)))
            for st in self._streams:
                cur.wait_stream(st)
            for o in outs:                                  # tensors were allocated on side streams
                for t in o:
                    if t is not None:
                        t.record_stream(cur)
        return tuple(None if o[0] is None else (o[0] if len(o) == 1 else torch.cat(o)) for o in zip(*outs))

    # ------------------------------------------------------------------ small ops
    def linear(self, x, weight, bias, relu=False):
        _require_gpu(x, 'linear input')
        x = x.to(torch.float32).contiguous()
        weight = weight.detach().to(self.device, torch.float32).contiguous()
        bias = bias.detach().to(self.device, torch.float32).contiguous() if bias is not None else None
        y = torch.empty((x.shape[0], weight.shape[0]), dtype=torch.float32, device=self.device)
        native.check(self.lib.wsi_linear(_ptr(x), _ptr(weight), _ptr(bias), _ptr(y), x.shape[0], x.shape[1],
                                         weight.shape[0], int(relu), _stream()), 'wsi_linear')
        return y


class AutoTrunkEngine:
    """TrunkEngine that picks its precision mode per checkpoint AND data, so a caller cannot silently leave the 1e-3 logit
    contract (BASELINE.json north_star).  mx (fp16 + MX-fp6 cross terms) is ~1.35x faster than parity (bf16x2 split); both
    hold the contract on every reference-generated weight / input family (tests/test_gpu_margin.py: mx <= 6.3e-4, parity
    <= 4.0e-4 at |logit| = 16), but every finite-precision error grows with the logit magnitude, so the choice is guarded:
      * at load: the folded weights must be representable (finite, inside the fp16 range) or mx is refused outright;
      * per slide (forward_tiles) / per head (forward_f32): `probe` images - a STRATIFIED sample over the whole tile list,
        never its first tiles: a slide's first raster tiles are one corner of the tissue - run in BOTH modes; |mx - parity|
        on the head logits (without a head: on the pooled features relative to their size) decides: mx if <= `tol`
        (default 5e-4: half the contract), parity otherwise;
      * with several ranks the decision must be ONE decision: callers that shard a slide (slide.infer_slide_cls) take the
        local `probe_tiles` error, all-reduce its maximum and hand it to `decide`, so no rank mixes modes into a gathered map.
    The decision and the measured value are kept in `.report` (utils.eval.predict_tumorbed returns it per slide)."""

    def __init__(self, state_dict, device, head=None, tol=5e-4, probe=32, **kw):
        self._kw = dict(kw)
        self._sd, self._dev, self.tol, self.probe = state_dict, device, float(tol), int(probe)
        self.report = {'mode': None, 'reason': 'not probed yet', 'probe_error': None}
        self._par = TrunkEngine(state_dict, device, planes=PARITY, head=head, **kw)
        self._mx = None
        reason = self._static_check(state_dict)
        if reason is None:
            self._mx = TrunkEngine(state_dict, device, planes=MX, head=head, **kw)
        else:
            self.report = {'mode': 'parity', 'reason': reason, 'probe_error': None}
        self._chosen = None if self._mx is not None else self._par
        self._slide_key, self._slide_ref, self._probed = None, None, None

    @staticmethod
    def _static_check(sd):
        """mx needs every BN-folded conv weight finite and well inside the fp16 range."""
        for key, w in sd.items():
            if not key.endswith('.weight') or getattr(w, 'dim', lambda: 0)() != 4:
                continue
            bn = key.replace('conv1.weight', 'bn1.weight').replace('conv2.weight', 'bn2.weight').replace('downsample.0.weight', 'downsample.1.weight')
            scale = 1.0
            if bn != key and bn in sd:
                var = sd[bn.replace('.weight', '.running_var')].detach().to('cpu', torch.float64)
                scale = (sd[bn].detach().to('cpu', torch.float64) / torch.sqrt(var + BN_EPS)).abs().max().item()
            m = float(w.detach().abs().max()) * scale
            if not np.isfinite(m) or m > 3.0e4:
                return 'folded weights of %s reach %.3g (outside the fp16 range of the mx mode)' % (key, m)
        return None

    # the TrunkEngine surface
    @property
    def planes(self):
        return (self._chosen or self._mx or self._par).planes

    @property
    def head_k(self):
        return self._par.head_k

    @property
    def lut(self):
        return self._par.lut

    def set_head(self, head):
        self._par.set_head(head)
        if self._mx is not None:
            self._mx.set_head(head)
            self._chosen = None                               # a new head changes the logit scale: probe again

    def linear(self, *a, **k):
        return self._par.linear(*a, **k)

    def reset(self):
        """Forget the decision: the next forward (or probe_tiles / decide pair) probes again."""
        if self._mx is not None:
            self._chosen, self._slide_key, self._slide_ref = None, None, None

    def _key_of(self, slide_u8, slide_id):
        """Identity of a slide for the one-decision-per-slide rule.  A caller-supplied `slide_id` (any hashable) wins; without one
        the slide is the TENSOR OBJECT the caller passes (held by weak reference), never its address: the caching allocator hands a
        new slide of the same shape the block the previous one just freed, and a (data_ptr, shape) key would then silently reuse the
        previous slide's mode without a probe."""
        if slide_id is not None:
            return ('id', slide_id)
        return ('obj', id(slide_u8))

    def _same_slide(self, slide_u8, key):
        if key != self._slide_key:
            return False
        if key[0] == 'id':
            return True
        ref = self._slide_ref
        return ref is not None and ref() is slide_u8            # a dead or different object with a recycled id() is a new slide

    def _stratified(self, total):
        n = min(self.probe, int(total))
        return torch.linspace(0, max(int(total) - 1, 0), n, device=self._dev).round().long()

    def _probe(self, run):
        """run(engine) -> (feat, logits, fmap) on the probe images.  Returns (error, what, n): |mx - parity| as described above."""
        fm, lm, _ = run(self._mx)
        fp, lp, _ = run(self._par)
        if lm is not None:
            return float((lm - lp).abs().max()), 'max |logit_mx - logit_parity|', int(lp.shape[0])
        return (float((fm - fp).abs().max() / fp.abs().max().clamp_min(1e-30)) * 2.0, '2 x max |feat_mx - feat_parity| / max |feat|',
                int(fp.shape[0]))

    def probe_tiles(self, slide_u8, tile_xy, ph, pw, slide_id=None):
        """Local probe error of mx on a stratified sample of `tile_xy` (0.0 without tiles or without an mx engine); the caller
        all-reduces the maximum over its ranks and calls decide().  The slide probed here is the slide the following decide()
        fixes the mode for: forward_tiles on it (same tensor object, or same `slide_id`) does not probe again."""
        self._probed = (self._key_of(slide_u8, slide_id), weakref.ref(slide_u8))
        if self._mx is None or int(tile_xy.shape[0]) == 0:
            return 0.0
        xy = tile_xy[self._stratified(tile_xy.shape[0])].contiguous()
        err, self._what, self._n = self._probe(lambda e: e.forward_tiles(slide_u8, xy, ph, pw, feat=True, logits=bool(e.head_k)))
        return err

    def probe_f32(self, x):
        """probe_tiles for the tensor-input path: local probe error on a stratified sample of the batch x (N, 3, H, W)."""
        if self._mx is None or int(x.shape[0]) == 0:
            return 0.0
        idx = self._stratified(x.shape[0])
        err, self._what, self._n = self._probe(lambda e: e.forward_f32(x[idx], feat=True, logits=bool(e.head_k)))
        return err

    def decide(self, err, scope='slide'):
        """Fix the mode for the coming forwards from a (rank-global) probe error.  After a probe_tiles() the decision belongs to
        THAT slide (r03 advisor finding: decide() used to leave the previous slide's key in place, so the first forward_tiles of
        every slide but the first probed again locally and could override the all-reduced decision on some ranks only)."""
        probed, self._probed = getattr(self, '_probed', None), None
        if probed is not None:
            self._slide_key, self._slide_ref = probed
        if self._mx is None:
            return
        ok = bool(np.isfinite(err)) and err <= self.tol
        self._chosen = self._mx if ok else self._par
        self.report = {'mode': 'mx' if ok else 'parity', 'probe_error': float(err), 'scope': scope,
                       'reason': '%s = %.2e %s tol %.1e on %d stratified probe images' %
                                 (getattr(self, '_what', 'probe error'), err, '<=' if ok else '>', self.tol, getattr(self, '_n', 0))}

    def forward_tiles_verified(self, slide_u8, tile_xy, ph, pw, reduce_max=None):
        """One slide (or one rank's shard of it), mx FIRST and checked afterwards: the whole tile list runs in mx, the stratified sample of
        probe_tiles once more in parity, and max |logit_mx - logit_parity| on the sample - reduced over the ranks by `reduce_max` (a
        collective: every rank must call, shards without tiles contribute 0) - decides whether the mx logits stand or the list is run
        again in parity.  Same sample, same rule and same report as probe_tiles + decide, but the mx forward of the sample and the host
        synchronisation BEFORE the slide's main pass are gone (r05: the drop-in API ran 3-4 % behind the bare engine); the price is a
        second pass over the slide in the rare case the answer is 'parity'.  Needs a head (logits).  Returns (feat=None, logits, None)."""
        n = int(tile_xy.shape[0])
        if self._mx is None:                                  # (the static weight check refused mx: report says why)
            if reduce_max is not None:
                reduce_max(0.0)                               # the other ranks' collective
            self.report['order'] = 'parity only'
            return self._par.forward_tiles(slide_u8, tile_xy, ph, pw, logits=True) if n else \
                (None, torch.zeros((0, self.head_k), dtype=torch.float32, device=tile_xy.device), None)
        out = self._mx.forward_tiles(slide_u8, tile_xy, ph, pw, logits=True) if n else None
        err, nprobe = 0.0, 0
        if n:
            idx = self._stratified(n)
            lp = self._par.forward_tiles(slide_u8, tile_xy[idx].contiguous(), ph, pw, logits=True)[1]
            err, nprobe = float((out[1][idx] - lp).abs().max()), int(idx.shape[0])
        if reduce_max is not None:
            err = float(reduce_max(err))
        self._what, self._n = 'max |logit_mx - logit_parity|', nprobe
        self._probed = None
        self.decide(err)
        self.report['order'] = 'mx first, verified on the sample afterwards'
        if self._chosen is self._par and n:
            out = self._par.forward_tiles(slide_u8, tile_xy, ph, pw, logits=True)
        if out is None:
            out = (None, torch.zeros((0, self.head_k), dtype=torch.float32, device=tile_xy.device), None)
        return out

    def forward_f32(self, x, feat=False, logits=False, fmap=False, tap=None):
        if self._chosen is None:
            self.decide(self.probe_f32(x), scope='head')
        return self._chosen.forward_f32(x, feat=feat, logits=logits, fmap=fmap, tap=tap)

    def forward_tiles(self, slide_u8, tile_xy, ph, pw, feat=False, logits=True, fmap=False, tap=None, slide_id=None):
        # one decision per slide: a slide this engine has not decided for (or a reset) probes first; chunks of the same slide - the
        # same tensor object or the same caller-supplied slide_id - reuse the decision, including one made by probe_tiles + decide
        key = self._key_of(slide_u8, slide_id)
        if self._mx is not None and (self._chosen is None or not self._same_slide(slide_u8, key)):
            self.decide(self.probe_tiles(slide_u8, tile_xy, ph, pw, slide_id=slide_id))
        return self._chosen.forward_tiles(slide_u8, tile_xy, ph, pw, feat=feat, logits=logits, fmap=fmap, tap=tap)


# ---------------------------------------------------------------------- standalone device ops
def pf_pack(x, planes):
    """f32 NCHW GPU tensor -> zero-initialised padded-flat buffer (uint8 tensor)."""
    lib = native.load()
    _require_gpu(x, 'pf_pack input')
    x = x.to(torch.float32).contiguous()
    n, c, h, w = x.shape
    buf = torch.zeros(lib.wsi_pf_bytes(n, h, w, c, planes), dtype=torch.uint8, device=x.device)
    native.check(lib.wsi_pf_pack(_ptr(x), _ptr(buf), n, c, h, w, planes, _stream()), 'wsi_pf_pack')
    return buf


def pf_unpack(buf, n, c, h, w, planes):
    lib = native.load()
    _require_gpu(buf, 'pf_unpack input')
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=buf.device)
    native.check(lib.wsi_pf_unpack(_ptr(buf), _ptr(out), n, c, h, w, planes, _stream()), 'wsi_pf_unpack')
    return out


def pf_zeros(n, c, h, w, planes, device):
    lib = native.load()
    return torch.zeros(lib.wsi_pf_bytes(n, h, w, c, planes), dtype=torch.uint8, device=device)


def prepack_conv(weight, bn, planes, device):
    """weight (cout,cin,k,k) fp32; bn = (gamma, beta, mean, var) or None -> (wpk, bias) device tensors."""
    lib = native.load()
    w = np.ascontiguousarray(torch.as_tensor(weight).detach().cpu().to(torch.float32).numpy())
    cout, cin, k = w.shape[0], w.shape[1], w.shape[2]
    nbytes = lib.wsi_prepack_conv_bytes(cout, cin, k, planes)
    if nbytes == 0:
        raise ValueError('unsupported conv shape %s' % (w.shape,))
    pk = np.empty(nbytes, np.uint8)
    bias = np.empty(cout, np.float32)
    if bn is None:
        args = [None] * 4
        keep = []
    else:
        keep = [np.ascontiguousarray(torch.as_tensor(t).detach().cpu().to(torch.float32).numpy()) for t in bn]
        args = [_np_ptr(a) for a in keep]
    native.check(lib.wsi_prepack_conv(_np_ptr(w), *args, BN_EPS, cout, cin, k, planes, _np_ptr(pk), _np_ptr(bias)),
                 'wsi_prepack_conv')
    return torch.from_numpy(pk).to(device), torch.from_numpy(bias).to(device)


def conv_bn_act(x_pf, n, h, w, cin, cout, wpk, bias, stride=1, ksize=3, resid_pf=None, relu=True, planes=2):
    lib = native.load()
    out = pf_zeros(n, cout, h // stride, w // stride, planes, x_pf.device)
    if ksize == 3:
        rc = lib.wsi_conv3x3_bn_act(_ptr(x_pf), _ptr(out), _ptr(resid_pf), _ptr(wpk), _ptr(bias), n, h, w, cin, cout, stride,
                                    int(relu), planes, _stream())
    else:
        rc = lib.wsi_conv1x1_bn(_ptr(x_pf), _ptr(out), _ptr(wpk), _ptr(bias), n, h, w, cin, cout, stride, planes, _stream())
    native.check(rc, 'wsi_conv')
    return out


def tile_gather(slide_u8, tile_xy, ph, pw, lut):
    """(SH,SW,3) u8 slide + (N,2) int32 corners -> normalised (N,3,ph,pw) fp32 (reference
    utils/dataset.py:171-185 with the eval transform)."""
    lib = native.load()
    _require_gpu(slide_u8, 'slide')
    tile_xy = tile_xy.to(slide_u8.device, torch.int32).contiguous()
    n = tile_xy.shape[0]
    out = torch.empty((n, 3, ph, pw), dtype=torch.float32, device=slide_u8.device)
    native.check(lib.wsi_tile_gather(_ptr(slide_u8), slide_u8.stride(0), slide_u8.shape[0], slide_u8.shape[1],
                                     _ptr(tile_xy), _ptr(lut), _ptr(out), n, ph, pw, _stream()), 'wsi_tile_gather')
    return out


def stitch_add(pred, tile_logits, map_xy, dy, dx):
    """pred (C,MH,MW) float64 GPU += per-tile logits over dy x dx footprints at map_xy (T,2) int32."""
    lib = native.load()
    _require_gpu(pred, 'prediction map')
    if pred.dtype != torch.float64 or not pred.is_contiguous():
        raise ValueError('pred must be a contiguous float64 tensor')
    tile_logits = tile_logits.to(torch.float32).contiguous()
    map_xy = map_xy.to(pred.device, torch.int32).contiguous()
    t, c = tile_logits.shape
    native.check(lib.wsi_stitch_add(_ptr(tile_logits), _ptr(map_xy), t, c, dy, dx, _ptr(pred), pred.shape[1], pred.shape[2],
                                    _stream()), 'wsi_stitch_add')
    return pred


def exponent_span(values):
    """Device tensor of 2 ints: {smallest, largest} biased fp32 exponent of the nonzero finite values (stitch guard)."""
    lib = native.load()
    _require_gpu(values, 'values')
    v = values.to(torch.float32).contiguous()
    out = torch.empty(2, dtype=torch.int32, device=v.device)
    if v.numel() == 0:
        out[0], out[1] = 255, 0
        return out
    native.check(lib.wsi_exponent_span(_ptr(v), v.numel(), _ptr(out), _stream()), 'wsi_exponent_span')
    return out


def stitch_is_exact(span, addends_per_pixel):
    """True when float64 sums of the guarded fp32 values are provably exact, hence independent of the order of the tiles and
    of how ranks split them: exponent span + log2(addends per pixel) <= 29 (a 53-bit significand holds every partial sum).
    Outside the bound the map is still correct to float64 rounding (<= 2^-52 relative per addition), just not bit-reproducible."""
    lo, hi = (int(v) for v in span.cpu())
    return not (hi >= lo and (hi - lo) + int(np.ceil(np.log2(max(1, addends_per_pixel)))) > 29)


def check_stitch_exact(span, addends_per_pixel):
    """Raise if float64 sums of the guarded fp32 values could be inexact (order-dependent): see wsi_exponent_span."""
    lo, hi = (int(v) for v in span.cpu())
    if not stitch_is_exact(span, addends_per_pixel):
        raise RuntimeError('stitch: the per-tile values span 2^%d with up to %d addends per pixel: float64 sums are no longer exact, '
                           'so the accumulated map would depend on the order of the tiles' % (hi - lo, addends_per_pixel))


def stitch_add_dense(pred, tile_pred, map_xy):
    """pred (C,MH,MW) float64 GPU += tile_pred (T,C,ph,pw) fp32 blocks at map_xy (T,2) int32 (clipped at the border)."""
    lib = native.load()
    _require_gpu(pred, 'prediction map')
    if pred.dtype != torch.float64 or not pred.is_contiguous():
        raise ValueError('pred must be a contiguous float64 tensor')
    tile_pred = tile_pred.to(pred.device, torch.float32).contiguous()
    map_xy = map_xy.to(pred.device, torch.int32).contiguous()
    t, c, ph, pw = tile_pred.shape
    native.check(lib.wsi_stitch_add_dense(_ptr(tile_pred), _ptr(map_xy), t, c, ph, pw, _ptr(pred), pred.shape[1], pred.shape[2],
                                          _stream()), 'wsi_stitch_add_dense')
    return pred


def resize_nearest(x, size):
    """F.interpolate(x, size) in its default 'nearest' mode on a (B,C,H,W) fp32 GPU tensor (reference utils/eval.py:202-206)."""
    lib = native.load()
    _require_gpu(x, 'tensor')
    x = x.to(torch.float32).contiguous()
    b, c, h, w = x.shape
    out = torch.empty((b, c, int(size[0]), int(size[1])), dtype=torch.float32, device=x.device)
    native.check(lib.wsi_resize_nearest_f32(_ptr(x), b * c, h, w, _ptr(out), out.shape[2], out.shape[3], _stream()), 'wsi_resize_nearest_f32')
    return out


def softmax_threshold_argmax(pred, class_probs, mask=None, heat_mode=None, want_probs=True):
    """pred (C,H,W) float64 GPU -> (classes u8 (H,W), probs f64 (C,H,W) or None, heat u8 (H,W) or None)."""
    lib = native.load()
    _require_gpu(pred, 'prediction map')
    pred = pred.contiguous()
    c, h, w = pred.shape
    th = torch.tensor([float(v) for v in class_probs][:c], dtype=torch.float64, device=pred.device)
    probs = torch.empty_like(pred) if want_probs else None
    classes = torch.empty((h, w), dtype=torch.uint8, device=pred.device)
    heat = torch.empty((h, w), dtype=torch.uint8, device=pred.device) if heat_mode is not None else None
    m = mask.to(pred.device, torch.uint8).contiguous() if mask is not None else None
    native.check(lib.wsi_softmax_threshold_argmax(_ptr(pred), c, h * w, _ptr(th), _ptr(probs), _ptr(classes), _ptr(m),
                                                  0 if heat_mode in (None, 'cls') else 1, _ptr(heat), _stream()),
                 'wsi_softmax_threshold_argmax')
    return classes, probs, heat
