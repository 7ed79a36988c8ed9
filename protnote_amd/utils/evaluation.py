"""Evaluation metrics on the device: the twins of the reference's torcheval BinaryAUPRC / MultilabelAUPRC and
BinaryBinnedAUPRC / MultilabelBinnedAUPRC (ProtNoteTrainer.py:477-485).  Scores stay in HBM for the whole evaluation
(no per-batch D2H as at ProtNoteTrainer.py:540-543); the arithmetic is in csrc/metrics.hip.  There is no host-side
implementation in this package - the numpy restatement used to check the kernels lives in oracle/metrics_oracle.py
(test infrastructure).

Labels without a positive: their AP is 0/0.  `empty_label_ap` (default 0.0) is the value such a label contributes to the
macro mean - torcheval's MultilabelAUPRC(average="macro") averages over ALL num_labels and scores an all-negative
label 0 (its curve code replaces the NaN recalls by 1.0), which is why the reference masks unrepresented labels
(`only_represented_labels`, ProtNoteTrainer.py:469-472,517-519).  torcheval is not installed here, so this convention
is "parity unpinned"; pass `empty_label_ap=None` to leave those labels out of the mean instead.  `ap_per_label`
always reports them as NaN."""
import torch

from .. import _lib


def _macro(ap: torch.Tensor, empty_label_ap) -> float:
    if empty_label_ap is None:
        return float(torch.nanmean(ap))
    return float(torch.where(torch.isnan(ap), torch.full_like(ap, float(empty_label_ap)), ap).mean())


_LABEL_KIND = {torch.float32: 0, torch.int64: 1, torch.uint8: 2, torch.bool: 2}


def _label_kind(labels: torch.Tensor) -> int:
    if labels.dtype not in _LABEL_KIND:
        raise TypeError(f"labels must be float32 / int64 / uint8 / bool, got {labels.dtype}")
    return _LABEL_KIND[labels.dtype]


def _require_hip_device(device):
    if torch.device(device).type != "cuda":
        raise RuntimeError("protnote_amd metrics run on an MI355X HIP device only (no CPU fallback); "
                           f"got device {device!r}")


class DeviceAveragePrecision:
    """Exact mAP on the device: the twin of torcheval `BinaryAUPRC` (micro, over all flattened pairs) +
    `MultilabelAUPRC` (per label, macro-averaged) as the reference uses them with ESTIMATE_MAP: False
    (ProtNoteTrainer.py:477-479, :540-543, :655-657), without moving a single score to the host.

    update(scores [B, N_L] f32, labels [B, N_L]) transposes the batch into a label-major accumulator
    (4 + 1 bytes per pair: 16 GB for 100 k sequences x 32 k labels); compute() sorts every label's column
    (rocPRIM segmented radix sort) and evaluates AP = sum over distinct thresholds of (recall step) x precision in
    f64.  Labels without a positive: NaN in `ap_per_label`, `empty_label_ap` in the macro mean (module docstring)."""

    def __init__(self, num_labels: int, capacity: int, device, growable: bool = False, empty_label_ap=0.0):
        _require_hip_device(device)
        self.empty_label_ap = empty_label_ap
        self.num_labels, self.capacity, self.n, self.growable = int(num_labels), max(int(capacity), 1), 0, growable
        self.keys = torch.empty(self.num_labels, self.capacity, dtype=torch.int32, device=device)
        self.hits = torch.empty(self.num_labels, self.capacity, dtype=torch.uint8, device=device)

    def reset(self):
        self.n = 0

    def update(self, scores: torch.Tensor, labels: torch.Tensor):
        _lib.require_hip(scores, labels)
        B, NL = scores.shape
        if NL != self.num_labels or labels.shape != scores.shape:
            raise ValueError(f"expected [B, {self.num_labels}] scores and labels, got {tuple(scores.shape)}, "
                             f"{tuple(labels.shape)}")
        if self.n + B > self.capacity:
            if not self.growable:
                raise RuntimeError(f"DeviceAveragePrecision: capacity {self.capacity} exceeded ({self.n} + {B})")
            cap = max(2 * self.capacity, self.n + B)
            keys = torch.empty(self.num_labels, cap, dtype=torch.int32, device=self.keys.device)
            hits = torch.empty(self.num_labels, cap, dtype=torch.uint8, device=self.keys.device)
            keys[:, :self.n], hits[:, :self.n] = self.keys[:, :self.n], self.hits[:, :self.n]
            self.keys, self.hits, self.capacity = keys, hits, cap
        scores = scores.float().contiguous()
        labels = labels.contiguous()
        _lib.check(_lib.lib().pn_ap_append(_lib.ptr(scores), NL, _lib.ptr(labels), _label_kind(labels), NL, B, NL,
                                           _lib.ptr(self.keys), _lib.ptr(self.hits), self.capacity, self.n,
                                           _lib.stream_ptr()))
        self.n += B

    # ------------------------------------------------------------------------------------------------------------
    # multi-GPU: the evaluation set is sharded by proteins (each rank scored its own sequences against all labels).
    # Nothing is gathered onto every GPU.  Per-label AP: labels are dealt to the ranks in contiguous blocks and ONE
    # all-to-all moves each label's columns to its owner (5 B per pair of the whole set, 1/W of it per GPU).  Micro AP:
    # a sample sort - the global histogram of the keys' upper 16 bits (one 512 KB all-reduce) cuts the key space into W
    # ranges of about equal population, a second all-to-all sends every pair to the owner of its range, each owner ranks
    # its range (pn_ap_partial, given the positives / pairs of the higher ranges) and one all-reduce adds the partials.
    # Equal keys share a range, so tie groups never straddle GPUs: the result is the single-GPU arithmetic.
    # ------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _exchange(chunks):
        """Variable-size all-to-all of 1-D tensors: chunks[v] goes to rank v; returns what every rank sent here."""
        import torch.distributed as dist

        W, rank, dev = dist.get_world_size(), dist.get_rank(), chunks[0].device
        cnt = torch.tensor([c.numel() for c in chunks], dtype=torch.int64, device=dev)
        allc = [torch.empty_like(cnt) for _ in range(W)]
        dist.all_gather(allc, cnt)
        recv = [int(allc[v][rank]) for v in range(W)]
        sent = [int(c.numel()) for c in chunks]
        inp = torch.cat(chunks)
        out = torch.empty(sum(recv), dtype=inp.dtype, device=dev)
        if dist.get_backend() == "nccl":  # RCCL over xGMI
            dist.all_to_all_single(out, inp, recv, sent)
        else:  # gloo (single-GPU dry runs and the CPU-transport tests): staged through the host
            o = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(o, inp.cpu(), recv, sent)
            out.copy_(o)
        return list(out.split(recv))

    def _compute_sharded(self, micro: bool) -> dict:
        import torch.distributed as dist

        L, W, rank, dev = _lib.lib(), dist.get_world_size(), dist.get_rank(), self.keys.device
        nt = torch.tensor([self.n], dtype=torch.int64, device=dev)
        ns = [torch.empty_like(nt) for _ in range(W)]
        dist.all_gather(ns, nt)
        ns = [int(v) for v in ns]
        n_tot = sum(ns)
        if n_tot == 0:
            raise RuntimeError("DeviceAveragePrecision.compute() before any update()")
        NL = self.num_labels
        k_loc, h_loc = self.keys[:, :self.n], self.hits[:, :self.n]
        # ---- per-label AP: rank v ranks labels [bounds[v], bounds[v+1]) over the proteins of ALL ranks
        bounds = [(NL * v) // W for v in range(W + 1)]
        rk = self._exchange([k_loc[bounds[v]:bounds[v + 1]].reshape(-1) for v in range(W)])
        rh = self._exchange([h_loc[bounds[v]:bounds[v + 1]].reshape(-1) for v in range(W)])
        mine = bounds[rank + 1] - bounds[rank]
        lmax = max(bounds[v + 1] - bounds[v] for v in range(W))
        ap_me = torch.full((lmax,), float("nan"), dtype=torch.float64, device=dev)
        np_me = torch.zeros(lmax, dtype=torch.int64, device=dev)
        if mine > 0:
            keys = torch.cat([t.view(mine, ns[u]) for u, t in enumerate(rk)], dim=1).contiguous()
            hits = torch.cat([t.view(mine, ns[u]) for u, t in enumerate(rh)], dim=1).contiguous()
            nbytes = L.pn_ap_ws_bytes(mine, n_tot, n_tot, 0)
            if nbytes == 0:
                _lib.check(1)
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(L.pn_ap_compute(_lib.ptr(keys), _lib.ptr(hits), mine, n_tot, n_tot, _lib.ptr(ap_me), _lib.ptr(np_me),
                                       None, None, _lib.ptr(ws), nbytes, _lib.stream_ptr()))
            del ws, keys, hits
        del rk, rh
        ap_all = [torch.empty_like(ap_me) for _ in range(W)]
        np_all = [torch.empty_like(np_me) for _ in range(W)]
        dist.all_gather(ap_all, ap_me)
        dist.all_gather(np_all, np_me)
        ap = torch.cat([a[:bounds[v + 1] - bounds[v]] for v, a in enumerate(ap_all)])
        npos = torch.cat([a[:bounds[v + 1] - bounds[v]] for v, a in enumerate(np_all)])
        out = {"ap_per_label": ap, "positives_per_label": npos, "map_macro": _macro(ap, self.empty_label_ap)}
        if not micro:
            return out
        # ---- micro AP: sample sort by the upper 16 key bits (u32 keys live in int32 storage)
        # (memory: beside the 5 B / pair accumulator this holds 4 B (hi16, transient) + 1 B (dest) + 1 B (one mask at a time)
        #  + the 5 B / pair send buffers - not W masks and int64 temporaries)
        kf, hf = k_loc.reshape(-1), h_loc.reshape(-1)
        hi16 = torch.bitwise_and(torch.bitwise_right_shift(kf, 16), 0xFFFF)  # int32, like the keys' storage
        hist = torch.bincount(hi16, minlength=65536)
        dist.all_reduce(hist)
        top = hist.flip(0)  # descending key order: range 0 holds the best-ranked pairs
        before = torch.cumsum(top, 0) - top
        lut = torch.clamp((before * W) // torch.clamp(top.sum(), min=1), max=W - 1).flip(0)
        lut = lut.to(torch.uint8 if W <= 255 else torch.int32)
        dest = lut[hi16]
        del hi16
        send_k, send_h = [], []
        for v in range(W):
            m = dest == v
            send_k.append(kf[m])
            send_h.append(hf[m])
            del m
        del dest
        mk = torch.cat(self._exchange(send_k))
        mh = torch.cat(self._exchange(send_h))
        del send_k, send_h
        tot = torch.stack([mh.sum(dtype=torch.int64), torch.tensor(mk.numel(), dtype=torch.int64, device=dev)])
        tots = [torch.empty_like(tot) for _ in range(W)]
        dist.all_gather(tots, tot)
        tp_before = sum(int(t[0]) for t in tots[:rank])
        k_before = sum(int(t[1]) for t in tots[:rank])
        part = torch.zeros(1, dtype=torch.float64, device=dev)
        pnp = torch.zeros(1, dtype=torch.int64, device=dev)
        m = int(mk.numel())
        nbytes = L.pn_ap_partial_ws_bytes(m)
        if nbytes == 0:
            _lib.check(1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.pn_ap_partial(_lib.ptr(mk) if m else None, _lib.ptr(mh) if m else None, m, tp_before, k_before,
                                   _lib.ptr(part), _lib.ptr(pnp), _lib.ptr(ws), nbytes, _lib.stream_ptr()))
        both = torch.stack([part[0], pnp[0].double()])
        dist.all_reduce(both)
        out["map_micro"] = float(both[0] / both[1]) if float(both[1]) > 0 else float("nan")
        return out

    def compute(self, micro: bool = True) -> dict:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return self._compute_sharded(micro)
        keys, hits, n, cap = self.keys, self.hits, self.n, self.capacity
        if n == 0:
            raise RuntimeError("DeviceAveragePrecision.compute() before any update()")
        L = _lib.lib()
        dev = keys.device
        ap = torch.empty(self.num_labels, dtype=torch.float64, device=dev)
        npos = torch.empty(self.num_labels, dtype=torch.int64, device=dev)
        mic = torch.empty(1, dtype=torch.float64, device=dev) if micro else None
        mic_n = torch.empty(1, dtype=torch.int64, device=dev) if micro else None
        nbytes = L.pn_ap_ws_bytes(self.num_labels, n, cap, int(micro))
        if nbytes == 0:
            _lib.check(1)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.pn_ap_compute(_lib.ptr(keys), _lib.ptr(hits), self.num_labels, n, cap, _lib.ptr(ap), _lib.ptr(npos),
                                   _lib.ptr(mic) if micro else None, _lib.ptr(mic_n) if micro else None,
                                   _lib.ptr(ws), nbytes, _lib.stream_ptr()))
        out = {"ap_per_label": ap, "positives_per_label": npos, "map_macro": _macro(ap, self.empty_label_ap)}
        if micro:
            out["map_micro"] = float(mic)
        del ws
        return out


class DeviceBinnedAUPRC:
    """Streaming binned AUPRC, the twin of torcheval `BinaryBinnedAUPRC(threshold=T)` +
    `MultilabelBinnedAUPRC(num_labels, threshold=T)` (ESTIMATE_MAP: True, ProtNoteTrainer.py:481-485): thresholds
    `linspace(0, 1, T)`, a prediction counts at threshold t when `p >= t`.  State: two [(N_L+1), T+1] u64 histograms
    (the last row pools all labels); multi-GPU = one all-reduce of the histograms."""

    def __init__(self, num_labels: int, device, threshold=50, empty_label_ap=0.0):
        _require_hip_device(device)
        self.empty_label_ap = empty_label_ap
        thr = torch.linspace(0, 1.0, threshold) if isinstance(threshold, int) else torch.as_tensor(threshold)
        self.thresholds = thr.float().to(device).contiguous()
        self.T, self.num_labels = int(self.thresholds.numel()), int(num_labels)
        self.pos = torch.zeros(self.num_labels + 1, self.T + 1, dtype=torch.int64, device=device)
        self.all = torch.zeros_like(self.pos)

    def reset(self):
        self.pos.zero_()
        self.all.zero_()

    def update(self, scores: torch.Tensor, labels: torch.Tensor):
        _lib.require_hip(scores, labels)
        B, NL = scores.shape
        if NL != self.num_labels or labels.shape != scores.shape:
            raise ValueError(f"expected [B, {self.num_labels}] scores and labels")
        scores, labels = scores.float().contiguous(), labels.contiguous()
        _lib.check(_lib.lib().pn_binned_hist_update(_lib.ptr(scores), NL, _lib.ptr(labels), _label_kind(labels), NL, B,
                                                    NL, _lib.ptr(self.thresholds), self.T, _lib.ptr(self.pos),
                                                    _lib.ptr(self.all), _lib.stream_ptr()))

    def compute(self) -> dict:
        import torch.distributed as dist

        pos, cnt = self.pos, self.all
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            both = torch.stack([pos, cnt])
            dist.all_reduce(both)
            pos, cnt = both[0].contiguous(), both[1].contiguous()
        out = torch.empty(self.num_labels + 1, dtype=torch.float64, device=pos.device)
        npos = torch.empty(self.num_labels + 1, dtype=torch.int64, device=pos.device)
        _lib.check(_lib.lib().pn_binned_auprc(_lib.ptr(pos), _lib.ptr(cnt), self.num_labels, self.T, _lib.ptr(out),
                                              _lib.ptr(npos), 1, _lib.stream_ptr()))
        per_label = out[:self.num_labels]
        return {"ap_per_label": per_label, "positives_per_label": npos[:self.num_labels],
                "map_macro": _macro(per_label, self.empty_label_ap), "map_micro": float(out[self.num_labels])}
