"""Host-side helpers around the Kaldi-style data interface (scp files, per-utterance npz / wav): cheap frame
counts for length-balanced sharding, without decoding the payloads."""
import os
import sys
import time
import wave
import zipfile

from numpy.lib import format as npformat


def host_threads(limit=4):
  """Cap torch's CPU thread pool in a process whose arithmetic runs on the GPU.  torch sizes the pool by the HOST's core
  count (128 on the MI355X boxes) even when the process may use 16 of them; every host thread that touches a tensor (the
  staging thread's pinned copies, a writer pool) then spawns its own pool of spinning workers and starves the loader's
  processes -- measured: 95 ms instead of 8 ms to stage one batch (tools/loader_probe.py)."""
  import torch
  try:
    cpus = len(os.sched_getaffinity(0))
  except AttributeError:
    cpus = os.cpu_count() or 1
  torch.set_num_threads(max(1, min(limit, cpus)))


def npz_frames(path, key="mix"):
    """Frame count T of the (257, T) array `key` in a feats npz (steps/extract_feats.py:90 writes zlib-compressed
    npz): only the .npy header of the zip member is inflated."""
    with zipfile.ZipFile(path) as z:
        with z.open(key + ".npy") as f:
            version = npformat.read_magic(f)
            if version == (1, 0):
                shape, _, _ = npformat.read_array_header_1_0(f)
            else:
                shape, _, _ = npformat.read_array_header_2_0(f)
    return int(shape[1]) if len(shape) > 1 else int(shape[0])


def wav_frames(path, hop=128):
    """STFT frame count 1 + N // hop of a wav file, from its header."""
    with wave.open(path, "rb") as w:
        return 1 + w.getnframes() // hop


def features_from_pcm(pcm, dev):
  """The on-GPU feature front end of a wav batch.  pcm = the arch's WavCollator batch: {'flat': int16 tensor
  holding every signal of the batch, key-major ('mix', 'source1', ...), longest utterance first; 'keys'; 'lens': samples per
  utterance} -> (mix (R,F), [source (R,F)...] packed rows, their Packing): STFT magnitudes by sk_stft into the (T,B,F) grid,
  then the valid rows.  One H2D copy for the whole batch; everything is enqueued on the CURRENT stream."""
  import torch
  from . import ops
  from .packing import Packing
  ns = [int(n) for n in pcm['lens']]
  B, F = len(ns), 257
  pk = Packing.from_lens([1 + n // 128 for n in ns], dev)
  flat = pcm['flat']
  if flat.device != torch.device(dev):
    flat = (flat if flat.is_pinned() else flat.pin_memory()).to(dev, non_blocking=True)
  feats, at, total = [], 0, sum(ns)
  for _ in pcm['keys']:
    out = torch.zeros(pk.T, B, F, device=dev)
    ops.stft_batch(flat[at:at + total], lengths=ns, out=out, out_offs=[b * F for b in range(B)],
                   stride_t=[B * F] * B, stride_f=[1] * B)
    at += total
    feats.append(pk.pack(out))
  flat.record_stream(torch.cuda.current_stream(dev))
  return feats[0], feats[1:], pk


# ----------------------------------------------------------------------------------------------- staging ahead of the step
class Prefetcher:
  """Iterates a DataLoader of the arch's batches and hands them over ALREADY ON THE GPU, as packed rows.

  The reference's loop (steps/train_qsub.py:113-122 over archs/uPIT.py:66-79,160-167) inflates the npz files, packs
  and copies every batch to the GPU synchronously in front of the step that consumes it.  At 36 ms per step that host
  work is the bound, so here a background thread takes the batches from the loader (whose workers inflate and pack in
  parallel), stages them through pinned memory and copies them on its own HIP stream, `depth` batches ahead:
    * PackedSequence batches (TrainSet): PackedSequence.data IS the engine's row layout -- one pinned copy + one
      asynchronous H2D per key, no padding anywhere; the batch arrives as {'packed': (mix, [sources], Packing)};
    * PCM batches (WavTrainSet, --wav-input): the int16 samples are copied and the STFT runs on the copy stream too.
  The consumer's stream waits for the batch's event; nothing on the host blocks.  Everything else in a batch (names,
  ...) passes through untouched."""

  _END = object()

  def __init__(self, loader, device, depth=2):
    import queue
    import threading
    self.loader, self.device, self.depth = loader, device, max(1, int(depth))
    self._queue_mod, self._threading = queue, threading
    self._stuck = None          # a staging thread that did not end when its consumer left early

  def __len__(self):
    return len(self.loader)

  def __iter__(self):
    import torch
    if self._stuck is not None:
      if self._stuck.is_alive():
        raise RuntimeError("Prefetcher: the staging thread of an earlier, abandoned pass is still inside the loader; "
                           "two threads must not share one loader iterator")
      self._stuck = None
    # The loader's iterator is made HERE, in the consumer's thread (with persistent workers: the worker processes are
    # started -- forked -- from this thread, not from a side thread of a process that holds the GPU, and a new pass
    # resets the same iterator only after the previous pass's thread is known to have left it).
    it = iter(self.loader)
    q = self._queue_mod.Queue(maxsize=self.depth)
    stop = self._threading.Event()
    dev = torch.device(self.device)
    stream = torch.cuda.Stream(device=dev)

    def put(item):
      while not stop.is_set():
        try:
          q.put(item, timeout=0.1)
          return True
        except self._queue_mod.Full:
          continue
      return False

    timing = os.environ.get("SEPKERN_PREFETCH_TIMING") == "1"      # diagnostic: where the staging thread's time goes
    acc = {"loader": 0.0, "stage": 0.0, "queue": 0.0, "n": 0}

    def work():
      try:
        torch.cuda.set_device(dev)
        with torch.cuda.stream(stream):
          while True:
            t0 = time.perf_counter()
            try:
              batch = next(it)
            except StopIteration:
              break
            t1 = time.perf_counter()
            staged = self.stage(batch, dev)
            ev = torch.cuda.Event()
            ev.record(stream)
            if timing:
              ev.synchronize()
            t2 = time.perf_counter()
            if not put((staged, ev)):
              return
            t3 = time.perf_counter()
            acc["loader"] += t1 - t0; acc["stage"] += t2 - t1; acc["queue"] += t3 - t2; acc["n"] += 1
        put(self._END)
        if timing and acc["n"]:
          print("prefetch: per batch %.1f ms waiting for the loader, %.1f ms staging (incl. the copies), %.1f ms waiting for "
                "the consumer" % tuple(1e3 * acc[k] / acc["n"] for k in ("loader", "stage", "queue")), file=sys.stderr, flush=True)
      except BaseException as e:          # re-raised in the consumer
        put(e)

    th = self._threading.Thread(target=work, name="sepkern-prefetch", daemon=True)
    th.start()
    try:
      while True:
        item = q.get()
        if item is self._END:
          break
        if isinstance(item, BaseException):
          raise item
        staged, ev = item
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(ev)
        for t in self._tensors(staged):
          t.record_stream(cur)             # allocated on the copy stream's pool, consumed on this one
        yield staged
    finally:
      # (normal end: the thread has put _END and is gone.  Early exit -- an exception in the step, a break: it may be
      # blocked in next(it); it returns as soon as the loader hands over that batch, finds `stop` set and leaves.)
      stop.set()
      th.join(timeout=60)
      if th.is_alive():
        self._stuck = th

  @staticmethod
  def _tensors(staged):
    p = staged.get('packed') if isinstance(staged, dict) else None      # (other batch types pass through as they came)
    if p is not None:
      yield p[0]
      for s in p[1]:
        yield s
      yield p[2].lens              # (lens / offs / perm are views of one staging tensor)

  @staticmethod
  def stage(batch, dev):
    """One batch -> {'packed': (mix (R,F), [source (R,F)...], Packing), <other keys unchanged>} on `dev`, enqueued on the
    CURRENT stream."""
    import torch
    from torch.nn.utils.rnn import PackedSequence
    from .packing import Packing
    if not isinstance(batch, dict):
      return batch
    if 'pcm' in batch:                   # WavCollator: {'pcm': {'flat': int16 tensor, 'keys', 'lens'}}: one pinned copy, STFT here
      pcm = batch['pcm']
      host = pcm['flat'] if pcm['flat'].is_pinned() else pcm['flat'].pin_memory()
      mix, sources, pk = features_from_pcm(dict(pcm, flat=host), dev)
      out = {k: v for k, v in batch.items() if k != 'pcm'}
      out['packed'] = (mix, sources, pk)
      out['_keepalive'] = host
      return out
    seqs = {k: v for k, v in batch.items() if isinstance(v, PackedSequence)}
    if 'mix' not in seqs:
      return batch
    pk = Packing.from_batch_sizes(seqs['mix'].batch_sizes, dev)
    order = ['mix'] + sorted((k for k in seqs if k.startswith('source')), key=lambda k: int(k[6:]))
    hosts, devs = [], []
    for k in order:
      h = seqs[k].data.pin_memory()
      hosts.append(h)
      devs.append(h.to(dev, non_blocking=True))
    out = {k: v for k, v in batch.items() if k not in seqs}
    out['packed'] = (devs[0], devs[1:], pk)
    out['_keepalive'] = hosts               # pinned staging must outlive the asynchronous copies
    return out
