"""Training / evaluation loop -- same entry point and flags as the reference's train.py
(--one_iter --datasets --multiplier --device --weight_decay --clip_grad_norm --dropout
--learning_rate --num_epochs --projection_size), plus --synthetic / --steps / --dtype / --root_map / --manifest_root / --ibm_root / --resume.

What changes against train.py:42-304:
  * features come from the fused HIP frontend on the GPU (data.collate_fn), not DataLoader workers;
  * log_softmax + transpose + CTCLoss (train.py:69-77) is one fused call (ops.ctc_loss) on the
    model's [B,T',V] logits -- same value and gradient, no transposed copy;
  * bf16 activations instead of fp16 autocast + GradScaler (no loss scaling needed), so the
    reference's clip-before-unscale quirk (train.py:79-83) disappears;
  * Adam state in fp32 over one flat arena (two launches per step: norm, clip+Adam) instead of
    bitsandbytes Adam8bit;
  * data-parallel training: one process per GPU, gradients all-reduced in ~25 MB buckets of the
    flat gradient arena on RCCL, launched from grad-ready hooks so they overlap the rest of
    backward (the reference is single-GPU).
Launch N ranks with ``python -m torch.distributed.run --nproc-per-node N -m speech_amd.train ...``.
"""
import argparse
import ctypes
import os
import time

import torch
import torch.distributed as dist

from . import data, decoder, net, ops
from ._lib import ADAM8_BLOCK, SUMSQ_WORDS, check, lib, ptr, stream_ptr


def lr_lambda_linear_warmup(current_step, num_warmup_steps, num_training_steps):
    """train.py:17-24: linear warm-up then linear decay to zero."""
    if current_step < num_warmup_steps:
        return float(current_step) / float(max(1, num_warmup_steps))
    return max(0.0, float(num_training_steps - current_step)
               / float(max(1, num_training_steps - num_warmup_steps)))


def get_linear_schedule_with_warmup(optimizer, num_warmup_steps, num_training_steps, last_epoch=-1):
    return torch.optim.lr_scheduler.LambdaLR(
        optimizer, lambda s: lr_lambda_linear_warmup(s, num_warmup_steps, num_training_steps), last_epoch)


class IterMeter(object):
    """keeps track of total iterations (train.py:29-39)"""

    def __init__(self):
        self.val = 0

    def step(self):
        self.val += 1

    def get(self):
        return self.val


class FlatAdam(torch.optim.Optimizer):
    """Adam over one flat fp32 arena.  Every parameter's .data and .grad become views into two
    contiguous buffers (16-byte aligned slices), so the global grad norm, the clip and the update
    are two kernel launches, and gradient buckets for the all-reduce are plain slices.
    state_bits = 8 keeps the moments as bitsandbytes-style block-wise 8-bit codes (train.py:254 Adam8bit;
    blocks of 2048 elements, one absmax per block and moment, every parameter aligned to a block boundary so
    that no block spans two tensors): 2 bytes of state per parameter instead of 8, and 16 instead of 28 bytes
    of traffic per parameter and step.  One stated difference to bitsandbytes is left: it keeps 32-bit moments
    for tensors below 4096 elements (biases, LayerNorm / BatchNorm vectors, weight_c); here those are quantised
    too, each in blocks of its own."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decoupled_weight_decay=True, state_bits=32):
        params = [p for p in params if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if state_bits not in (8, 32):
            raise ValueError("state_bits must be 32 or 8")
        self.plist = params
        dev = params[0].device
        # 8-bit state: every parameter starts on a quantisation-block boundary, so that no block (one absmax per
        # 2048 elements and moment) spans two tensors -- bitsandbytes quantises each tensor on its own
        align = ADAM8_BLOCK if state_bits == 8 else 4
        self.offsets, total = [], 0
        for p in params:
            self.offsets.append(total)
            total += (p.numel() + align - 1) // align * align
        self.total = total
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.state_bits = state_bits
        if state_bits == 32:
            self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
            self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        else:
            nblk = (total + ADAM8_BLOCK - 1) // ADAM8_BLOCK
            self.exp_avg = torch.full((total,), 127, dtype=torch.uint8, device=dev)      # code 127 = 0 (signed table)
            self.exp_avg_sq = torch.zeros(total, dtype=torch.uint8, device=dev)           # code 0 = 0 (unsigned table)
            self.absmax1 = torch.zeros(nblk, dtype=torch.float32, device=dev)
            self.absmax2 = torch.zeros(nblk, dtype=torch.float32, device=dev)
        self.sumsq = torch.zeros(SUMSQ_WORDS, dtype=torch.float32, device=dev)   # [0] = sum, rest partials
        for p, off in zip(params, self.offsets):
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1).float())
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.flat_grad[off:off + n].view(p.shape)
        self.t = 0
        self.decoupled = decoupled_weight_decay

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        for p, off in zip(self.plist, self.offsets):  # autograd may have replaced a view
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    @torch.no_grad()
    def step(self, max_norm=0.0, grad_div=1.0):
        g = self.param_groups[0]
        self.t += 1
        with torch.cuda.device(self.flat.device):
            s = stream_ptr()
            sumsq = None
            if max_norm and max_norm > 0:
                check(lib().sp_sumsq_f32(ptr(self.flat_grad), self.total, ptr(self.sumsq), s), "sp_sumsq_f32")
                sumsq = self.sumsq
            if self.state_bits == 8:
                check(lib().sp_adam8_step_f32(ptr(self.flat), ptr(self.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                              ptr(self.absmax1), ptr(self.absmax2), self.total, float(g["lr"]),
                                              float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                              float(g["weight_decay"]), self.t, int(self.decoupled), ptr(sumsq),
                                              float(max_norm or 0.0), float(grad_div), s), "sp_adam8_step_f32")
                return
            check(lib().sp_adam_step_f32(ptr(self.flat), ptr(self.flat_grad), ptr(self.exp_avg),
                                         ptr(self.exp_avg_sq), self.total, float(g["lr"]),
                                         float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                                         float(g["weight_decay"]), self.t, int(self.decoupled), ptr(sumsq),
                                         float(max_norm or 0.0), float(grad_div), s), "sp_adam_step_f32")

    def state_dict(self):
        sd = {"t": self.t, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
              "state_bits": self.state_bits,
              "param_groups": [{k: v for k, v in self.param_groups[0].items() if k != "params"}]}
        if self.state_bits == 8:
            sd["absmax1"], sd["absmax2"] = self.absmax1.clone(), self.absmax2.clone()
        return sd

    def load_state_dict(self, sd):
        if sd.get("state_bits", 32) != self.state_bits:
            raise ValueError(f"checkpoint holds {sd.get('state_bits', 32)}-bit optimizer state, this optimizer {self.state_bits}-bit")
        self.t = sd["t"]
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        if self.state_bits == 8:
            self.absmax1.copy_(sd["absmax1"])
            self.absmax2.copy_(sd["absmax2"])


def _host_staged(group, t):
    """gloo never sees a device tensor here (see BucketedAllReduce)."""
    return dist.get_backend(group) == "gloo" and t.is_cuda


def all_reduce_sum(t, group=None, op=None):
    """Blocking all-reduce of `t` in place on either backend: RCCL takes the device tensor; gloo gets a host copy."""
    op = dist.ReduceOp.SUM if op is None else op
    if _host_staged(group, t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def broadcast_(t, src=0, group=None):
    if _host_staged(group, t):
        h = t.detach().cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)
    return t


class BucketedAllReduce:
    """Data-parallel gradient exchange: the flat gradient arena is cut, from its END (the
    classifier and the last SRU layers, whose gradients are ready first), into buckets of about
    ``bucket_mb``; a post-accumulate hook on every parameter counts its bucket down and the bucket's
    all-reduce (sum; the mean's 1/world is folded into the optimizer's grad_div) is launched
    asynchronously on RCCL as soon as the bucket is complete, overlapping the rest of backward.

    RCCL: the collective is enqueued on RCCL's own stream behind an event on the launch stream; the host never
    waits (`finish` makes the launch stream wait on the collectives' events), so a bucket may be launched from the
    autograd thread in the middle of backward.
    gloo (CPU tests; the two-ranks-on-one-card rehearsal): device tensors are NOT handed to ProcessGroupGloo.  Its
    device-tensor path stages through pinned memory on side streams and runs `hipStreamSynchronize` and copies from
    its own worker threads while the autograd thread is still launching backward kernels; round 2's intermittent
    two-rank hang sat in exactly that combination.  A completed bucket is therefore only RECORDED
    during backward (same hook logic, same order) and exchanged in `finish`, on the calling thread, through a host
    buffer: no foreign thread touches the HIP runtime and every rank enters the collectives in the recorded order."""

    def __init__(self, opt, group=None, bucket_mb=25.0):
        self.opt, self.group = opt, group
        self.world = dist.get_world_size(group)
        self.host_staged = _host_staged(group, opt.flat_grad)
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []   # [lo, hi, n_params]
        self.bucket_of = {}
        hi, count, lo = opt.total, 0, opt.total
        for idx in range(len(opt.plist) - 1, -1, -1):
            lo = opt.offsets[idx]
            count += 1
            self.bucket_of[idx] = len(self.buckets)
            if hi - lo >= cap or idx == 0:
                self.buckets.append([lo, hi, count])
                hi, count = lo, 0
        self.pending = [b[2] for b in self.buckets]
        self.handles = []
        self.order = []          # buckets in the order they were launched this step (the same on every rank)
        self.done = set()
        self.hooks = [self._make_hook(idx) for idx in range(len(opt.plist))]
        # keyed by storage address: inside autograd functions a parameter may arrive as another Python
        # wrapper of the same tensor
        self.index_of = {p.data_ptr(): idx for idx, p in enumerate(opt.plist)}
        for idx, p in enumerate(opt.plist):
            p.register_post_accumulate_grad_hook(self.hooks[idx])
        if self.host_staged:
            self.stage = torch.empty(max(hi - lo for lo, hi, _ in self.buckets), dtype=torch.float32).pin_memory()
        # start from identical replicas
        broadcast_(opt.flat, 0, group)

    def _make_hook(self, idx):
        def hook(_param):
            # idempotent within a step: a gradient written straight into the arena is reported by
            # ops.GradSink, and autograd may still run the parameter's hook with an empty gradient
            if idx in self.done:
                return
            self.done.add(idx)
            b = self.bucket_of[idx]
            self.pending[b] -= 1
            if self.pending[b] == 0:
                self._launch(b)
        return hook

    def _launch(self, b):
        self.order.append(b)
        if self.host_staged:
            return           # exchanged in finish(), in this order
        lo, hi, _ = self.buckets[b]
        self.handles.append(dist.all_reduce(self.opt.flat_grad[lo:hi], group=self.group, async_op=True))

    def param_ready(self, p):
        """For gradients a backward kernel wrote straight into the arena (ops.GradSink): autograd's
        post-accumulate hook does not fire for them."""
        idx = self.index_of.get(p.data_ptr())
        if idx is not None:
            self.hooks[idx](p)

    def finish(self):
        """Call after backward: launch whatever did not fire (unused parameters), wait for all."""
        for b, left in enumerate(self.pending):
            if left > 0:
                self._launch(b)
        if self.host_staged:
            for b in self.order:
                lo, hi, _ = self.buckets[b]
                h = self.stage[:hi - lo]
                h.copy_(self.opt.flat_grad[lo:hi])        # blocking: the launch stream has produced the bucket
                dist.all_reduce(h, group=self.group)
                self.opt.flat_grad[lo:hi].copy_(h)
        for h in self.handles:
            h.wait()
        self.handles = []
        self.order = []
        self.done = set()
        self.pending = [b[2] for b in self.buckets]


def train_step(model, optimizer, spectrograms, labels, label_lengths, clip_grad_norm=2.0, dp=None,
               max_label_len=None):
    """One optimisation step on a featurised batch (train.py:62-87).  Returns the loss tensor
    (device, no sync)."""
    optimizer.zero_grad()
    # FlatAdam: every .grad is a zeroed view of the flat arena, so accumulating backward kernels may
    # write straight into it (ops.GradSink) instead of through a temporary and an add per parameter
    sink = ops.GradSink(dp) if isinstance(optimizer, FlatAdam) else None
    ops.set_grad_sink(sink)
    try:
        if hasattr(model, "ctc_loss"):      # SRModel: classifier + CTC as one node, time-major logits
            loss = model.ctc_loss(spectrograms, labels, label_lengths, max_label_len)
        else:
            output = model(spectrograms)                                   # [B, T', V]
            B, Tp, _ = output.shape
            # train.py:72-74: every utterance is scored over all T' (padded) encoder frames
            input_lengths = torch.full((B,), Tp, dtype=torch.int32, device=output.device)
            loss = ops.ctc_loss(output.float(), labels, input_lengths, label_lengths, blank=0,
                                max_label_len=max_label_len)
        loss.backward()
    finally:
        ops.set_grad_sink(None)
    world = 1
    if dp is not None:
        dp.finish()
        world = dp.world
    optimizer.step(max_norm=clip_grad_norm, grad_div=float(world))
    return loss.detach()


def train(batch_size, model, train_loader, optimizer, scheduler, epoch, iter_meter, clip_grad_norm,
          dp=None, max_steps=None, log=print):
    model.train()
    data_len = len(train_loader.dataset) if hasattr(train_loader, "dataset") else len(train_loader)
    start = batch_start = time.time()
    for batch_idx, batch in enumerate(train_loader):
        spectrograms, labels, label_lengths = batch
        spectrograms = spectrograms.cuda()
        loss = train_step(model, optimizer, spectrograms, labels.cuda(), label_lengths.cuda(),
                          clip_grad_norm, dp)
        scheduler.step()
        iter_meter.step()
        if batch_idx % 100 == 0:
            t100 = round(time.time() - batch_start, 1)
            log("Train Epoch: {} [{}/{} ({:.0f}%)]\tLoss: {:.6f}\tT100B: {}".format(
                epoch, batch_idx, data_len, 100.0 * batch_idx / max(1, data_len), loss.item(), t100))
            batch_start = time.time()
        if max_steps is not None and batch_idx + 1 >= max_steps:
            break
    return round(time.time() - start)


@torch.no_grad()
def eval_dataset(model, loader, name, log=print, dtype=torch.float32, group=None, world=1):
    """train.py:105-144: mean loss, unweighted mean of per-utterance CER / WER.  The reference evaluates
    outside autocast, i.e. in fp32 (train.py:162-168): `dtype` is the activation type for this pass (None = the
    model's training type).  world > 1: every rank evaluates its own share of the loader (build_loaders shards
    them) and the sums are all-reduced, so no rank sits in a collective while another one evaluates alone."""
    model.eval()
    saved = getattr(model, "compute_dtype", None)
    if dtype is not None and saved is not None:
        model.compute_dtype = dtype
    eval_loss, eval_cer, eval_wer, n = 0.0, [], [], 0
    try:
        for batch in loader:
            spectrograms, labels, label_lengths = batch
            spectrograms, labels, label_lengths = spectrograms.cuda(), labels.cuda(), label_lengths.cuda()
            output = model(spectrograms).float()
            B, Tp, _ = output.shape
            input_lengths = torch.full((B,), Tp, dtype=torch.int32, device=output.device)
            loss, _, _ = ops.ctc_fwd_bwd(output, labels, input_lengths, label_lengths, want_grad=False)
            eval_loss += float(loss)
            n += 1
            preds, targets = decoder.greedy_decoder(output, labels, label_lengths)
            for tgt, pred in zip(targets, preds):
                eval_cer.append(decoder.cer(tgt, pred))
                eval_wer.append(decoder.wer(tgt, pred))
    finally:
        if saved is not None:
            model.compute_dtype = saved
    sums = torch.tensor([eval_loss, float(n), sum(eval_cer), sum(eval_wer), float(len(eval_cer))], dtype=torch.float64)
    if world > 1:
        sums = all_reduce_sum(sums.cuda() if torch.cuda.is_available() else sums, group).cpu()
    eval_loss, n, cer_sum, wer_sum, n_utt = (float(v) for v in sums)
    avg_cer = cer_sum / n_utt
    avg_wer = wer_sum / n_utt
    log("{}: Average loss: {:.4f}, Average CER: {:4f} Average WER: {:.4f}\n".format(
        name, eval_loss / max(1.0, n), avg_cer, avg_wer))
    return avg_cer


def save_checkpoint(path, epoch, model, optimizer, iterations=None):
    """train.py:171-178 key names (+ the scheduler's step count, which the reference's commented-out resume
    would have needed too)."""
    ck = {"epoch": epoch, "model_state_dict": model.state_dict(), "optimizer_state_dict": optimizer.state_dict()}
    if iterations is not None:
        ck["iterations"] = int(iterations)
    torch.save(ck, path)


def resume_from(path, model, optimizer=None, map_location="cuda"):
    """Load a checkpoint in either shape the reference writes.  With the full dict of save_checkpoint the
    optimizer's moments / step count come back too (train.py:249-260's resume stub loads both); returns
    (epoch, iterations) to continue from, (0, 0) for a bare state_dict."""
    ck = torch.load(path, map_location=map_location, weights_only=True)
    model.load_checkpoint(ck)
    epoch = iterations = 0
    if isinstance(ck, dict) and "model_state_dict" in ck:
        epoch = int(ck.get("epoch", 0))
        iterations = int(ck.get("iterations", 0))
        osd = ck.get("optimizer_state_dict")
        if optimizer is not None and osd is not None and hasattr(optimizer, "load_state_dict") and "exp_avg" in osd:
            optimizer.load_state_dict(osd)       # raises if the state widths differ (8-bit vs fp32)
    return epoch, iterations


def setup_distributed():
    """One process per GPU; RANK/LOCAL_RANK/WORLD_SIZE from the launcher."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if torch.cuda.is_available():
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        # ranks only ever wait for each other inside a step's exchange or the end-of-epoch sums (evaluation is
        # sharded), so a rank that stops answering for this long has died
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo",
                                timeout=datetime.timedelta(minutes=int(os.environ.get("SP_DIST_TIMEOUT_MIN", "30"))))
    return int(os.environ.get("RANK", "0")), world


# train.py:182-192: the reference's manifest list (relative to the working directory, like there)
TV_TRAIN_DATASETS_PATHS = [
    "datasets/first/sorted_train_cer_0.1.pkl",
    "datasets/second/sorted_train_cer_0.1.pkl",
    "datasets/third/sorted_train_cer_0.1.pkl",
    "datasets/fourth/sorted_train_cer_0.1.pkl",
    "datasets/fifth/sorted_train_cer_0.1.pkl",
    "datasets/sixth/sorted_train_cer_0.1.pkl",
    "datasets/gigaspeech/sorted_train_youtube_filtered.pkl",
    "datasets/gigaspeech/sorted_train_podcast_filtered.pkl",
    "datasets/gigaspeech/sorted_train_audiobook_filtered.pkl",
]
LIBRISPEECH_TRAIN_PATH = "datasets/librispeech/sorted_train_librispeech.pkl"


class RankBatchSampler(torch.utils.data.Sampler):
    """This rank's share of a SortedDataset's batch indices: one permutation per epoch that every rank derives
    from the same seed, cut to a multiple of the world size, so that all ranks take the SAME number of steps (a
    rank with one batch more would wait forever in its bucket all-reduces)."""

    def __init__(self, n_batches, rank=0, world=1, shuffle=True, seed=7):
        self.n, self.rank, self.world, self.shuffle, self.seed, self.epoch = n_batches, rank, world, shuffle, seed, 0

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        return self.n // self.world

    def __iter__(self):
        if self.shuffle:
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist()
        else:
            order = list(range(self.n))
        return iter(order[self.rank:len(self) * self.world:self.world])


class ShardSampler(torch.utils.data.Sampler):
    """Evaluation share of one rank: indices rank, rank + world, ... of an optional permutation every rank derives
    from the same seed.  Nothing is dropped or repeated (ranks may differ by one item: no collective runs inside an
    evaluation pass, the sums are exchanged once at its end)."""

    def __init__(self, n, rank=0, world=1, shuffle=False, seed=11):
        self.n, self.rank, self.world, self.shuffle, self.seed = n, rank, world, shuffle, seed

    def __len__(self):
        return len(range(self.rank, self.n, self.world))

    def __iter__(self):
        order = (torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed)).tolist()
                 if self.shuffle else list(range(self.n)))
        return iter(order[self.rank::self.world])


class FeaturizingLoader:
    """DataLoader workers read and decode the clips (file IO, as in the reference); the featuriser runs on the
    GPU, which forked workers must not touch, so collate_fn is applied here, in the training process."""

    def __init__(self, loader, data_type):
        self.loader, self.data_type = loader, data_type
        self.dataset = loader.dataset

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for clips in self.loader:
            yield data.collate_fn(clips, self.data_type)


def _clips(batch):
    return batch


def build_loaders(hparams, rank, world):
    """(train_loader, sampler, [(name, eval_loader), ...]) as train.py:202-239 builds them: CombinedTVLibriSpeech for
    training, SortedTV for the "tv" evaluation, IBMDataset for the "ibm" one whose CER picks the checkpoint."""
    bs = hparams["batch_size"]
    workers = hparams.get("num_workers", 6)
    if hparams["synthetic"]:
        train_set = data.SyntheticSorted(hparams["synthetic_clips"], bs, seed=7)
        tv_eval = data.SyntheticSorted(max(bs, hparams["synthetic_clips"] // 8), bs, seed=11, max_s=6.0)
        ibm_eval, workers = None, 0
    else:
        datasets = hparams["datasets"].split("-")
        assert "tv" in datasets and "libri" in datasets
        device = hparams.get("device") or 0
        root_map = hparams.get("root_map")
        prefix = hparams.get("manifest_root", "")
        tv_paths = [os.path.join(prefix, p) for p in TV_TRAIN_DATASETS_PATHS]
        if hparams.get("tv_manifests"):
            tv_paths = list(hparams["tv_manifests"])
        train_set = data.CombinedTVLibriSpeech(os.path.join(prefix, LIBRISPEECH_TRAIN_PATH), tv_paths, bs, device, root_map)
        # train.py:208-212 swaps "train" for "eval" in the manifest NAME (a prefix directory may say "train" too)
        tv_eval_paths = [os.path.join(os.path.dirname(p), os.path.basename(p).replace("train", "eval"))
                         for p in tv_paths if "gigaspeech" not in p]
        tv_eval = data.SortedTV(tv_eval_paths, bs, device, root_map)
        ibm_eval = data.IBMDataset(hparams.get("ibm_root", "/hd1/ibm"))
    sampler = RankBatchSampler(len(train_set), rank, world, shuffle=True)
    if hparams["steps"]:
        sampler.n = min(sampler.n, hparams["steps"] * world)
    train_loader = FeaturizingLoader(torch.utils.data.DataLoader(
        train_set, batch_size=None, sampler=sampler, collate_fn=_clips, num_workers=workers, pin_memory=False), "train")
    # train.py:222-239: the tv evaluation walks its batches shuffled, the ibm one in order; each rank takes every
    # world-th batch / utterance (world 1: exactly the reference's loaders)
    evals = [("tv", FeaturizingLoader(torch.utils.data.DataLoader(
        tv_eval, batch_size=None, sampler=ShardSampler(len(tv_eval), rank, world, shuffle=True), collate_fn=_clips,
        num_workers=min(workers, 4)), "valid"))]
    if ibm_eval is not None:
        evals.append(("ibm", FeaturizingLoader(torch.utils.data.DataLoader(
            ibm_eval, batch_size=32 * hparams["multiplier"] * 2, sampler=ShardSampler(len(ibm_eval), rank, world),
            collate_fn=_clips, num_workers=min(workers, 4)), "valid")))
    return train_loader, sampler, evals


def main(hparams):
    rank, world = setup_distributed()
    torch.manual_seed(7 + rank)
    log = print if rank == 0 else (lambda *a, **k: None)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[hparams["dtype"]]
    model = net.SRModel(hparams["n_rnn_layers"], hparams["rnn_dim"], hparams["n_vocab"], hparams["n_feats"],
                        hparams["dropout"], hparams["projection_size"], compute_dtype=dtype).cuda()
    log("Num Model Parameters", sum(p.nelement() for p in model.parameters()))
    optimizer = FlatAdam(model.parameters(), lr=hparams["learning_rate"], weight_decay=hparams["weight_decay"],
                         state_bits=hparams.get("optim_bits", 32))
    start_epoch = start_iter = 0
    if hparams.get("resume"):      # weights, and with a full checkpoint the optimizer state, epoch and step count
        start_epoch, start_iter = resume_from(hparams["resume"], model, optimizer)
    dp = BucketedAllReduce(optimizer) if world > 1 else None

    train_loader, sampler, evals = build_loaders(hparams, rank, world)
    scheduler = get_linear_schedule_with_warmup(optimizer, 7000 // hparams["multiplier"],
                                                hparams["epochs"] * len(train_loader))
    iter_meter, last_cer = IterMeter(), 2.0
    if start_iter:                 # a resumed run continues the warm-up / decay where the checkpoint left it
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            scheduler.step(start_iter)
        iter_meter.val = start_iter
    eval_dtype = {"fp32": torch.float32, "same": None}[hparams.get("eval_dtype", "fp32")]
    for epoch in range(start_epoch + 1, (1 if hparams["one_iter"] else hparams["epochs"]) + 1):
        sampler.set_epoch(epoch)
        train(hparams["batch_size"], model, train_loader, optimizer, scheduler, epoch, iter_meter,
              hparams["clip_grad_norm"], dp, log=log)
        # train.py:147-179: the last evaluation's CER (ibm when present) picks the checkpoint.  Every rank evaluates
        # its share and the sums are all-reduced: no rank waits in a collective while another evaluates alone.
        log("\nevaluating...")
        for name, loader in evals:
            avg_cer = eval_dataset(model, loader, name, log, eval_dtype, None, world)
        if rank == 0 and avg_cer < last_cer and hparams["checkpoint"]:
            save_checkpoint(hparams["checkpoint"], epoch, model, optimizer, iter_meter.get())
        last_cer = avg_cer
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return last_cer


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--one_iter", action="store_true")
    p.add_argument("--datasets", type=str, default="tv-libri")
    p.add_argument("--multiplier", type=int, default=2)
    p.add_argument("--device", type=int)
    p.add_argument("--weight_decay", type=float, default=0.001)
    p.add_argument("--clip_grad_norm", type=float, default=2.0)
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--learning_rate", type=float, default=3e-4)
    p.add_argument("--num_epochs", type=int, default=45)
    p.add_argument("--projection_size", type=int, default=0)
    # build-side additions
    p.add_argument("--synthetic", action="store_true")
    p.add_argument("--synthetic_clips", type=int, default=512)
    p.add_argument("--steps", type=int, default=0)
    p.add_argument("--dtype", choices=["bf16", "fp16", "fp32"], default="bf16")
    p.add_argument("--checkpoint", type=str, default="")
    p.add_argument("--resume", type=str, default="", help="checkpoint to start from (either shape the reference writes)")
    p.add_argument("--eval_dtype", choices=["fp32", "same"], default="fp32")
    p.add_argument("--root_map", type=str, default="",
                   help="old=new[,old=new...]: path prefixes rewritten in the manifests (the reference hard-codes /data -> /hd<device+1>)")
    p.add_argument("--manifest_root", type=str, default="", help="directory the datasets/... manifest paths are relative to")
    p.add_argument("--ibm_root", type=str, default="/hd1/ibm")
    p.add_argument("--num_workers", type=int, default=6)
    p.add_argument("--optim_bits", type=int, choices=[32, 8], default=32,
                   help="optimizer state: 32 = fp32 moments, 8 = block-wise 8-bit moments as bitsandbytes Adam8bit (train.py:254)")
    a = p.parse_args(argv)
    return {
        "datasets": a.datasets, "multiplier": a.multiplier, "batch_size": 32 * a.multiplier,
        "epochs": a.num_epochs, "learning_rate": a.learning_rate, "n_rnn_layers": 10, "rnn_dim": 512,
        "dropout": a.dropout, "n_vocab": 28, "n_feats": data.N_MELS, "weight_decay": a.weight_decay,
        "clip_grad_norm": a.clip_grad_norm, "one_iter": a.one_iter, "projection_size": a.projection_size,
        "synthetic": a.synthetic, "synthetic_clips": a.synthetic_clips, "steps": a.steps,
        "dtype": a.dtype, "checkpoint": a.checkpoint, "resume": a.resume, "eval_dtype": a.eval_dtype,
        "optim_bits": a.optim_bits, "device": a.device, "manifest_root": a.manifest_root, "ibm_root": a.ibm_root, "num_workers": a.num_workers,
        "root_map": tuple(tuple(kv.split("=", 1)) for kv in a.root_map.split(",") if kv) or None,
    }


if __name__ == "__main__":
    main(parse_args())
