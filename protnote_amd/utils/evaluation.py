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

    def _gathered(self):
        """Scores of all ranks side by side along the protein axis (every rank computes the same metric)."""
        import torch.distributed as dist

        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return self.keys, self.hits, self.n, self.capacity
        W = dist.get_world_size()
        ns = [torch.zeros(1, dtype=torch.int64, device=self.keys.device) for _ in range(W)]
        dist.all_gather(ns, torch.tensor([self.n], dtype=torch.int64, device=self.keys.device))
        ns = [int(v) for v in ns]
        nmax = max(ns)
        if nmax == 0:
            return self.keys, self.hits, 0, self.capacity
        k_loc = torch.zeros(self.num_labels, nmax, dtype=torch.int32, device=self.keys.device)
        h_loc = torch.zeros(self.num_labels, nmax, dtype=torch.uint8, device=self.keys.device)
        k_loc[:, :self.n], h_loc[:, :self.n] = self.keys[:, :self.n], self.hits[:, :self.n]
        k_all = [torch.empty_like(k_loc) for _ in range(W)]
        h_all = [torch.empty_like(h_loc) for _ in range(W)]
        dist.all_gather(k_all, k_loc)
        dist.all_gather(h_all, h_loc)
        keys = torch.cat([k[:, :m] for k, m in zip(k_all, ns)], dim=1).contiguous()
        hits = torch.cat([h[:, :m] for h, m in zip(h_all, ns)], dim=1).contiguous()
        return keys, hits, sum(ns), sum(ns)

    def compute(self, micro: bool = True) -> dict:
        keys, hits, n, cap = self._gathered()
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
