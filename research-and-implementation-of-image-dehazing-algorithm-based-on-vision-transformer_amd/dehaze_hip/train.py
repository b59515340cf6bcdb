"""Training-step machinery: flat-buffer AdamW on the HIP kernel, bucketed RCCL gradient all-reduce
(one process per GPU over xGMI - replaces the reference's nn.DataParallel, TR:97), synthetic haze
batches and the step body of My_train.py (TR:212-250).
"""
import bisect
import math
import os
import weakref

import torch
import torch.distributed as dist

from . import ops


# ----------------------------------------------------------------------------- flat AdamW (K12)
class FlatAdamW(torch.optim.Optimizer):
    """torch.optim.AdamW semantics (TR:90-92: lr 2e-4, betas (0.9,0.999), eps 1e-8, wd 0.02) with all LIVE
    parameters, their gradients and both moments living in four flat fp32 buffers, updated by ONE launch
    of dhz_adamw_step.  Parameters that never receive a gradient (the reference's 108 dead attn.qkv.* /
    attn.proj.* tensors) are skipped exactly like torch skips `grad is None` params.

    The flat layout is in REVERSE registration order so that gradient buckets (contiguous slices of the
    flat gradient) complete in backward order.  state_dict()/load_state_dict() speak torch.optim.AdamW's
    positional format, so reference checkpoints ('optimizer' entry, TR:296) round-trip.

    Flattening happens lazily at the first zero_grad()/step() - i.e. after the model has been moved to its
    device, mirroring the reference's order (optimizer built before .cuda(), TR:90-98).  Moving the model
    afterwards would detach the parameters from the flat buffers (don't).
    """

    def __init__(self, model_or_params, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.02, live=None):
        names = {}
        if isinstance(model_or_params, torch.nn.Module):
            params = list(model_or_params.parameters())
            names = {id(p): n for n, p in model_or_params.named_parameters()}
            if live is None and hasattr(model_or_params, "live_parameters"):
                live = [p for _, p in model_or_params.live_parameters()]
        else:
            params = list(model_or_params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._all = params
        # frozen parameters (utils.freeze / a partial fine-tune) stay out of the flat buffers: neither updated nor decayed
        self._live = [p for p in (live if live is not None else params) if p.requires_grad]
        self._names = names
        self._flat = None
        self._step = 0
        self.grad_scale = 1.0          # set to 1/world_size by the gradient reducer (SUM all-reduce)

    # -- flat buffers
    def _ensure_flat(self):
        if self._flat is not None:
            return
        order = self._qkv_adjacent(list(reversed(self._live)))
        dev = order[0].device
        # every parameter starts on a 32-byte boundary: the kernels read weights with 16-byte loads, and the bf16 shadow of this
        # buffer (config 4) must be 16-byte aligned too; the only sizes that are not multiples of 8 floats are the 225 x H bias
        # tables.  The pad floats stay zero in all four buffers (zero gradient -> zero moments -> zero update)
        n = sum((p.numel() + 7) // 8 * 8 for p in order)
        npad = n
        fp = torch.zeros(npad, device=dev, dtype=torch.float32)
        # behind the gradients: a zeroed scratch region for the accumulation targets of the backward pass that are NOT parameter gradients
        # in parameter layout (the resampling convolutions' weight gradients in GEMM layout, bias gradients of derived biases): zeroed by
        # the same launch as the gradients (zero_grad), handed out by ops.zeros_f32 - a dozen fill launches per step otherwise
        nscr = (sum(p.numel() for p in order if p.dim() == 4) + 65536 + 7) // 8 * 8
        fg_all = torch.zeros(npad + nscr, device=dev, dtype=torch.float32)
        fg = fg_all[:npad]
        self._offsets = {}
        off = 0
        for p in order:
            off = (off + 7) // 8 * 8
            k = p.numel()
            fp[off:off + k].copy_(p.data.reshape(-1))
            if p.grad is not None:
                fg[off:off + k].copy_(p.grad.reshape(-1))
            p.data = fp[off:off + k].view_as(p)
            p.grad = fg[off:off + k].view_as(p)
            self._offsets[id(p)] = (off, k)
            off += k
        self._flat = dict(p=fp, g=fg, g_all=fg_all, scratch=fg_all[npad:], m=torch.zeros_like(fp), v=torch.zeros_like(fp), n=n)

    def _qkv_adjacent(self, order):
        """Within every attention layer put the three projection weights back to back (Wq, Wk, Wv) and the three biases
        likewise: the packed [3C, C] operand of the QKV GEMMs is then a VIEW of the flat buffer (ops.cat_rows) instead of a
        torch.cat per block and pass.  The six tensors occupy the same six consecutive slots as before (registration order
        q.w q.b k.w k.b v.w v.b), so bucket boundaries in backward order are unchanged."""
        pos = {self._names.get(id(p), ""): i for i, p in enumerate(order)}
        for name, i in list(pos.items()):
            if not name.endswith("query_projection.weight"):
                continue
            pre = name[:-len("query_projection.weight")]
            six = [pre + f"{m}_projection.{t}" for t in ("weight", "bias") for m in ("query", "key", "value")]
            if not all(n in pos for n in six):
                continue
            slots = sorted(pos[n] for n in six)
            if slots != list(range(slots[0], slots[0] + 6)):
                continue
            for slot, n in zip(slots, six):
                order[slot] = next(p for p in self._live if self._names.get(id(p)) == n)
        return order

    @property
    def flat_grad(self):
        self._ensure_flat()
        return self._flat["g"]

    def param_slices(self):
        """[(param, offset, numel)] in flat-buffer order (reverse registration, Q/K/V grouped - _qkv_adjacent)."""
        self._ensure_flat()
        return sorted(((p,) + self._offsets[id(p)] for p in self._live), key=lambda t: t[1])

    def zero_grad(self, set_to_none=False):
        self._ensure_flat()
        self._flat["g_all"].zero_()                       # the gradients and the scratch region behind them: one launch
        ops.ZERO_SCRATCH = [self._flat["scratch"], 0, weakref.ref(self)]
        for p in self._all:
            if id(p) not in self._offsets:
                p.grad = None

    @torch.no_grad()
    def step(self, closure=None):
        self._ensure_flat()
        g = self.param_groups[0]
        self._step += 1
        f = self._flat
        shadowed = "p16" in f
        ops.adamw_step_(f["p"], f["g"], f["m"], f["v"], g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                        g["weight_decay"], self._step, self.grad_scale, p16=f["p16"] if shadowed else None)
        if shadowed and ops.BF16_SHADOW is not None and ops.BF16_SHADOW[0] is f["p"]:
            ops.refresh_bf16_shadow_t()          # (the AdamW kernel wrote the bf16 copy itself; its transposes: one launch)
        if "p3" in f and ops.SPLIT_SHADOW is not None and ops.SPLIT_SHADOW[0] is f["p"]:
            ops.refresh_split_shadow()           # the three bf16 planes of the updated parameters: one launch
        self._record_versions()

    def _record_versions(self):
        """the version counters of the parameters (views of the flat buffer with counters of their own) and of the flat buffer at the
        moment the derived copies are current: a counter moves when anything but the optimizer kernel writes a parameter in place (a
        loaded checkpoint, a landscape probe)"""
        if getattr(self, "_vtab", None) is None:
            sl = self.param_slices()
            self._vtab = ([off for _, off, _ in sl], [p for p, _, _ in sl])
        self._pver = [p._version for p in self._vtab[1]]
        self._fver = self._flat["p"]._version

    def region_current(self, off, n):
        """True when no parameter overlapping flat[off : off + n] was written (counter-visibly) since the copies were derived -
        the per-lookup guard of ops.split_planes / split_planes_t / bf16_copy / bf16_copy_t."""
        pv = getattr(self, "_pver", None)
        if pv is None or self._flat["p"]._version != self._fver:
            return False
        offs, ps = self._vtab
        i = max(bisect.bisect_right(offs, off) - 1, 0)
        end = off + n
        while i < len(offs) and offs[i] < end:
            if ps[i]._version != pv[i]:
                return False
            i += 1
        return True

    def sync_shadows(self, force=False):
        """Re-derive the bf16 copy / the split planes if a parameter was written since the last update (or unconditionally with
        force=True: after writes that bump no version counter - p.data.copy_(...), raw-pointer kernels).  Reached through
        ops.sync_shadows() (train_step, Uformer.forward) and from any stale lookup (ops._shadow_fresh)."""
        f = self._flat
        if f is None:
            return
        stale = force or getattr(self, "_pver", None) is None or f["p"]._version != self._fver \
            or any(p._version != v for p, v in zip(self._vtab[1], self._pver))
        if stale:
            if "p16" in f and ops.BF16_SHADOW is not None and ops.BF16_SHADOW[0] is f["p"]:
                ops.refresh_bf16_shadow()
            if "p3" in f and ops.SPLIT_SHADOW is not None and ops.SPLIT_SHADOW[0] is f["p"]:
                ops.refresh_split_shadow()
            self._record_versions()

    sync_bf16_shadow = sync_shadows               # (earlier name)

    def _matrix_table(self):
        """[(offset, rows, cols)] of the matrices whose TRANSPOSES the shadows also keep (for the backward-data GEMMs): every 2-D
        weight with dimensions in 32s; the adjacent Q / K / V weights of an attention layer as ONE packed [3C, C] matrix (what the
        packed backward-data GEMM multiplies with)."""
        slices = self.param_slices()
        mats, i = [], 0
        while i < len(slices):
            p, off, k = slices[i]
            name = self._names.get(id(p), "")
            if name.endswith("query_projection.weight") and i + 2 < len(slices) and p.dim() == 2:
                pk, pv = slices[i + 1][0], slices[i + 2][0]
                if self._names.get(id(pk), "").endswith("key_projection.weight") and self._names.get(id(pv), "").endswith(
                        "value_projection.weight") and slices[i + 1][1] == off + k and slices[i + 2][1] == off + 2 * k \
                        and pk.shape == p.shape == pv.shape and p.shape[0] % 32 == 0 and p.shape[1] % 32 == 0:
                    mats.append((off, 3 * p.shape[0], p.shape[1]))
                    i += 3
                    continue
            if p.dim() == 2 and p.shape[0] % 32 == 0 and p.shape[1] % 32 == 0 and off % 8 == 0:
                mats.append((off, p.shape[0], p.shape[1]))
            i += 1
        return mats

    def _matrix_desc(self, mats):
        """device table [nmat, 4] = (offset, rows, cols, first 32 x 32 tile) of dhz_split3_planes_t / dhz_bf16_transpose_batched, the
        lookup set, and the tile count"""
        rows, t0 = [], 0
        for off, R, Cc in mats:
            rows.append([off, R, Cc, t0])
            t0 += (R // 32) * (Cc // 32)
        return torch.tensor(rows, dtype=torch.int32, device=self._flat["p"].device), {(off, R, Cc) for off, R, Cc in mats}, t0

    def enable_split_shadow(self):
        """Keep the three bf16 truncation planes of the flat parameter buffer (the pre-split weight operand of the six-term
        GEMMs, csrc/split6_gemm.hip), refreshed after every update; ops.split_planes hands out views of them."""
        self._ensure_flat()
        f = self._flat
        if "p3" not in f:
            f["p3"] = torch.empty((3, f["p"].numel()), device=f["p"].device, dtype=torch.bfloat16)
            mats = self._matrix_table()
            if mats:
                f["p3t"] = torch.zeros((3, f["p"].numel()), device=f["p"].device, dtype=torch.bfloat16)
                f["p3t_desc"], f["p3t_index"], f["p3t_ntiles"] = self._matrix_desc(mats)
        ops.set_split_shadow(f["p"], f["p3"], f.get("p3t"), f.get("p3t_desc"), f.get("p3t_index"), f.get("p3t_ntiles", 0))
        ops.refresh_split_shadow()
        self._record_versions()
        ops.SHADOW_OWNER = weakref.ref(self)

    def enable_bf16_shadow(self):
        """Keep a bf16 copy of the flat parameter buffer, refreshed after every update, and let ops.bf16_copy hand out views
        of it (BASELINE config 4)."""
        self._ensure_flat()
        f = self._flat
        if "p16" not in f:
            f["p16"] = f["p"].to(torch.bfloat16)
            mats = self._matrix_table()
            if mats:                            # bf16 copies of the transposes: backward-data runs the forward kernel on them
                f["p16t"] = torch.zeros_like(f["p16"])
                f["p16t_desc"], f["p16t_index"], f["p16t_ntiles"] = self._matrix_desc(mats)
        ops.set_bf16_shadow(f["p"], f["p16"], f.get("p16t"), f.get("p16t_desc"), f.get("p16t_index"), f.get("p16t_ntiles", 0))
        ops.refresh_bf16_shadow()
        self._record_versions()
        ops.SHADOW_OWNER = weakref.ref(self)

    def __del__(self):
        # the process-global shadow must not pin the flat buffers of a discarded optimizer
        try:
            f = self._flat
            if f is not None and ops.BF16_SHADOW is not None and ops.BF16_SHADOW[0] is f["p"]:
                ops.set_bf16_shadow(None, None)
            if f is not None and ops.SPLIT_SHADOW is not None and ops.SPLIT_SHADOW[0] is f["p"]:
                ops.set_split_shadow(None, None)
            if ops.SHADOW_OWNER is not None and ops.SHADOW_OWNER() in (None, self):
                ops.SHADOW_OWNER = None
        except Exception:
            pass

    # -- torch.optim.AdamW-compatible (positional) state
    def state_dict(self):
        self._ensure_flat()
        index = {id(p): i for i, p in enumerate(self._all)}
        state = {}
        if self._step > 0:
            for p in self._live:
                off, k = self._offsets[id(p)]
                state[index[id(p)]] = {"step": torch.tensor(float(self._step)),
                                       "exp_avg": self._flat["m"][off:off + k].view_as(p).clone(),
                                       "exp_avg_sq": self._flat["v"][off:off + k].view_as(p).clone()}
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        groups[0]["params"] = list(range(len(self._all)))
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        self._ensure_flat()
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
        steps = set()
        for i, st in sd["state"].items():
            p = self._all[int(i)]
            if id(p) not in self._offsets:
                continue
            off, k = self._offsets[id(p)]
            self._flat["m"][off:off + k].copy_(st["exp_avg"].reshape(-1))
            self._flat["v"][off:off + k].copy_(st["exp_avg_sq"].reshape(-1))
            steps.add(int(float(st["step"])))
        if steps:
            self._step = max(steps)


# ----------------------------------------------------------------------------- C1: bucketed gradient all-reduce
class GradReducer:
    """DDP-style gradient exchange for one-process-per-GPU data parallelism.

    Gradients live in one flat buffer (FlatAdamW.flat_grad, or a private one when used stand-alone);
    it is cut into ~bucket_mb contiguous buckets in backward order.  A post-accumulate hook on every
    live parameter counts arrivals; when a bucket is complete its slice is all-reduced (SUM) asynchronously
    on the process group's own stream (RCCL over xGMI on GPUs, gloo in the CPU tests) while backward
    continues.  wait() joins the outstanding collectives; the 1/world_size factor is folded into the
    optimizer kernel (FlatAdamW.grad_scale) instead of a separate pass over the 82.5 MB gradient.
    Dead parameters never enter a bucket (find_unused_parameters-equivalent).
    """

    def __init__(self, optimizer=None, params=None, bucket_mb=25.0, group=None, overlap=True, reserve_cus=None):
        self.group = group
        self.overlap = overlap        # False (bench.py --no-overlap): every bucket's collective is launched by wait(), after backward
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # CUs left to the collective kernels: every persistent grid of the HIP library is sized "workgroups per CU x (CUs - k)" from
        # here on (dhz_set_reserved_cus; csrc/api.hip).  Only with more than one rank, only on a GPU; default DHZ_COMM_RESERVE_CUS or 0.
        # Its single-GPU cost is recorded in profiles/r05_reserve_cus.txt; bench.py reports it as exchange.reserve_cus.
        if reserve_cus is None:
            reserve_cus = int(os.environ.get("DHZ_COMM_RESERVE_CUS", "0"))
        self.reserve_cus = 0
        if self.world > 1 and reserve_cus > 0 and torch.cuda.is_available() and overlap:
            from . import _lib
            _lib.call("dhz_set_reserved_cus", int(reserve_cus))
            self.reserve_cus = int(reserve_cus)
        self.opt = optimizer
        if optimizer is not None:
            slices = optimizer.param_slices()
            self.flat = optimizer.flat_grad
            optimizer.grad_scale = 1.0 / self.world
        else:
            params = [p for p in reversed(list(params)) if p.requires_grad]
            n = sum(p.numel() for p in params)
            self.flat = torch.zeros(n, device=params[0].device, dtype=params[0].dtype)
            slices, off = [], 0
            for p in params:
                p.grad = self.flat[off:off + p.numel()].view_as(p)
                slices.append((p, off, p.numel()))
                off += p.numel()
        self._slices = slices
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []            # [lo, hi, n_params]
        self.bucket_of = {}
        lo, cnt = 0, 0
        for p, off, k in slices:
            self.bucket_of[id(p)] = len(self.buckets)
            cnt += 1
            if off + k - lo >= cap:
                self.buckets.append([lo, off + k, cnt])
                lo, cnt = off + k, 0
        if cnt:
            self.buckets.append([lo, slices[-1][1] + slices[-1][2], cnt])
        self._pending = [0] * len(self.buckets)
        self._handles = []
        self._hooks = []
        self._seen = set()            # parameters already announced in this backward pass
        self.calls = None             # diagnostics: set to {} to count announcements per parameter id
        self.timing = None            # diagnostics: set to [] and wait() appends a HIP-event pair around its joins (bench.py: exposed exchange)
        if self.world > 1:
            for p, _, _ in slices:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            ops.GRAD_READY = self._on_grad_inplace      # wgrad kernels accumulate in place, bypassing autograd hooks

    def close(self):
        """give the reserved compute units back (dhz_set_reserved_cus is process-global: single-rank work that follows in the same
        process would otherwise keep the shrunken grids)"""
        if getattr(self, "reserve_cus", 0):
            try:
                from . import _lib
                _lib.call("dhz_set_reserved_cus", 0)
            except Exception:
                pass
            self.reserve_cus = 0

    def __del__(self):
        self.close()

    def plan(self, world=None, link_gbs=153.0):
        """The exchange this reducer performs per step, without performing it: bucket byte ranges in launch (= backward)
        order and the ring all-reduce time they imply on xGMI.  A ring over N GPUs moves 2 (N - 1) / N of the payload over
        every GPU's slowest used link (one ~153 GB/s xGMI link per ring neighbour: MI355X_MICROARCH / the task's numbers);
        RCCL can stripe several rings over the 7 links, so this is the conservative single-ring figure."""
        world = world or self.world
        total = sum(hi - lo for lo, hi, _ in self.buckets) * 4
        factor = 2.0 * (world - 1) / world if world > 1 else 0.0
        return {"world": world, "payload_bytes": total,
                "buckets": [{"lo": lo, "hi": hi, "bytes": (hi - lo) * 4, "params": n} for lo, hi, n in self.buckets],
                "ring_time_ms": 1e3 * factor * total / (link_gbs * 1e9),
                "ring_time_ms_per_bucket": [1e3 * factor * (hi - lo) * 4 / (link_gbs * 1e9) for lo, hi, _ in self.buckets]}

    def _on_grad_inplace(self, p):
        if id(p) in self.bucket_of:
            self._on_grad(p)

    def _on_grad(self, p):
        # A parameter can be announced twice in one backward: by the in-place wgrad path (ops.GRAD_READY, right after
        # its kernel is enqueued) and again by autograd's post-accumulate hook, which recent PyTorch also fires for the
        # None gradient the fused autograd nodes return for such parameters.  Only the first one counts.
        if self.calls is not None:
            self.calls[id(p)] = self.calls.get(id(p), 0) + 1
        if id(p) in self._seen:
            return
        self._seen.add(id(p))
        b = self.bucket_of[id(p)]
        self._pending[b] += 1
        if self._pending[b] == self.buckets[b][2] and self.overlap:
            lo, hi, _ = self.buckets[b]
            self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        """Join all bucket all-reduces of this backward pass (call before optimizer.step())."""
        if self.world > 1:
            # buckets whose hooks did not all fire (a parameter unused this step) are reduced here
            for b, (lo, hi, n) in enumerate(self.buckets):
                if 0 < self._pending[b] < n or (self._pending[b] == 0 and n > 0) or (not self.overlap and n > 0):
                    self._handles.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group,
                                                         async_op=True))
            ev = None
            if self.timing is not None and torch.cuda.is_available() and self.flat.is_cuda:
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            for h in self._handles:
                h.wait()
            if ev is not None:
                # between the two records the compute stream does nothing but wait for the collectives: the EXPOSED exchange time
                ev[1].record()
                self.timing.append(ev)
        self._handles = []
        self._pending = [0] * len(self.buckets)
        self._seen = set()

    def average_(self):
        """Stand-alone use (no FlatAdamW): turn the summed gradients into the mean."""
        if self.world > 1:
            self.flat.div_(self.world)

    def zero_grad(self):
        """Stand-alone use: zero the flat gradient and re-bind every parameter's .grad to its slice (torch optimizers'
        zero_grad(set_to_none=True) would detach them from the buffer the buckets are cut from)."""
        self.flat.zero_()
        for p, off, k in self._slices:
            p.grad = self.flat[off:off + k].view_as(p)


# ----------------------------------------------------------------------------- data + step
def synthetic_batch(batch, ps=128, seed=1234, device="cpu"):
    """Synthetic haze pairs: gt ~ U[0,1); hazy = clamp(t*gt + (1-t)*A, 0, 1) with
    per-sample transmission t ~ U(0.3,0.9) and airlight A ~ U(0.6,1.0).  Returns (target, input_)."""
    g = torch.Generator().manual_seed(seed)
    h, w = (ps, ps) if isinstance(ps, int) else ps
    gt = torch.rand(batch, 3, h, w, generator=g)
    t = 0.3 + 0.6 * torch.rand(batch, 1, 1, 1, generator=g)
    A = 0.6 + 0.4 * torch.rand(batch, 1, 1, 1, generator=g)
    hazy = (t * gt + (1 - t) * A).clamp(0, 1)
    return gt.to(device), hazy.to(device)


class SideStream:
    """A second HIP stream, optionally restricted to a subset of the compute units (hipExtStreamCreateWithCUMask), for work that is
    independent of the main stream's critical path: train_step(side=...) puts the no-gradient VGG19 passes of the contrastive
    loss (ground truth and hazy input: known before the step starts) on it, beside the model's forward.  cus = 0: an ordinary
    stream (all CUs).  Measured on MI355X: profiles/r04_concurrency.txt."""

    def __init__(self, device, cus=0, first=0):
        import ctypes
        self.device = torch.device(device)
        self.cus = int(cus)
        self._hip = None
        if self.cus > 0:
            hip = ctypes.CDLL("libamdhip64.so")
            total = torch.cuda.get_device_properties(self.device).multi_processor_count
            words = (total + 31) // 32
            mask = (ctypes.c_uint32 * words)()
            for i in range(first, min(first + self.cus, total)):
                mask[i // 32] |= 1 << (i % 32)
            st = ctypes.c_void_p()
            with torch.cuda.device(self.device):
                rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
            if rc != 0 or not st.value:
                raise RuntimeError(f"hipExtStreamCreateWithCUMask failed with {rc}")
            self._hip, self._raw = hip, st
            self.stream = torch.cuda.ExternalStream(st.value, device=self.device)
        else:
            self.stream = torch.cuda.Stream(device=self.device)

    def __del__(self):
        try:
            if self._hip is not None:
                self._hip.hipStreamDestroy(self._raw)
        except Exception:
            pass


def train_step(model, char_loss, cr_loss, optimizer, reducer, input_, target, w_char=1.0, w_cr=1.0, side=None):
    """One optimisation step, the body of TR:212-250 in fp32: zero_grad -> restored = model(input_) ->
    clamp(0,1) -> w_char*Charbonnier + w_cr*Contrast -> backward (bucketed all-reduce overlapped) ->
    AdamW.  Returns (loss, loss_rec, loss_cr) as device scalars (no host sync here; the reference's
    per-step .item() calls, TR:250-254, are left to the caller's logging cadence).
    side: a SideStream - the contrastive loss' no-gradient feature passes (target, input_) run on it beside the model's forward."""
    if isinstance(optimizer, FlatAdamW):
        if getattr(model, "act_dtype", None) == torch.bfloat16:
            if ops.BF16_SHADOW is None or optimizer._flat is None or ops.BF16_SHADOW[0] is not optimizer._flat["p"]:
                optimizer.enable_bf16_shadow()
        elif ops.SPLIT_BF16 == 6:
            if ops.SPLIT_SHADOW is None or optimizer._flat is None or ops.SPLIT_SHADOW[0] is not optimizer._flat["p"]:
                optimizer.enable_split_shadow()
        optimizer.sync_shadows()               # parameters written outside step() (a loaded checkpoint, a landscape probe)
                                               # must reach the GEMMs' derived weight copies: one launch, only when that happened
    if cr_loss is not None and hasattr(cr_loss, "vgg") and hasattr(cr_loss.vgg, "feature_dtype"):
        # config 4: the frozen feature stack follows the model's activation type (autocast covers the loss in the reference)
        cr_loss.vgg.feature_dtype = getattr(model, "act_dtype", None) or torch.float32
    standalone = reducer is not None and reducer.opt is None          # torch optimizer + stand-alone reducer (--optimizer adam)
    if standalone:
        reducer.zero_grad()
    else:
        optimizer.zero_grad()
    pn_taps, pn_done = None, None
    use_cr = w_cr > 0 and cr_loss is not None
    if use_cr and side is not None and hasattr(cr_loss, "reference_taps"):
        main = torch.cuda.current_stream()
        side.stream.wait_stream(main)                                 # the batch is resident
        with torch.cuda.stream(side.stream):
            pn_taps = cr_loss.reference_taps(target, input_)
            pn_done = torch.cuda.Event()
            pn_done.record(side.stream)
    restored = model(input_)
    loss_rec, clamped = char_loss.forward_clamped(restored, target)
    loss = w_char * loss_rec if w_char > 0 else 0
    loss_cr = None
    if use_cr:
        if pn_taps is not None:
            torch.cuda.current_stream().wait_event(pn_done)
            for t in pn_taps:
                t.record_stream(torch.cuda.current_stream())          # allocated on the side stream, consumed here
            loss_cr, _, _ = cr_loss(clamped, target, input_, pn_taps=pn_taps)
        else:
            loss_cr, _, _ = cr_loss(clamped, target, input_)
        loss = loss + w_cr * loss_cr
    loss.backward()
    if reducer is not None:
        reducer.wait()
        if standalone:
            reducer.average_()
    optimizer.step()
    return loss.detach(), loss_rec.detach(), (loss_cr.detach() if loss_cr is not None else None)


def psnr(a, b):
    mse = torch.mean((a.double() - b.double()) ** 2).item()
    return 10.0 * math.log10(1.0 / max(mse, 1e-20))
