"""``DeviceLoader`` -- a persistent batch source for the train loops, sized for a 288 GB accelerator.

The reference's loops call ``next(iter(DataLoader(dataset, batch_size, shuffle=True, collate_fn=...)))`` on every
update (scripts/train_speech.py:59-61, train_rir.py:37-40, train_location.py:59-62): each call builds a new iterator,
draws a new permutation and ``torch.load``s a whole batch of ~1.2 MB files in the training thread.  At B=64 that is
~80 MB of file parsing per step, an order of magnitude longer than the MI355X step itself.

This class keeps the same contract at the call site -- ``next(iter(loader))`` returns the 6-tuple of
``spec_dataset_preprocessing`` -- but ``iter()`` returns the loader itself (one persistent stream of batches), and
the samples come from one of two places:

* **resident** (default whenever the cropped dataset fits the budget): every sample is read ONCE by a thread pool,
  cropped to 500 frames (samples shorter than that are dropped, the collate's rule) and parked in device memory
  -- 1.2 MB per sample, 24 GB for the 20k-sample set of train_location.py, a twelfth of the card.  A batch is then
  three ``index_select``s on the device: no file, host or PCIe work per step.
* **streaming** (datasets larger than the budget): persistent worker PROCESSES (``torch.load`` is pickle parsing
  and holds the GIL, so threads do not scale) load and collate batches ahead into pinned memory, a feeder thread
  queues their copies on a side stream, and ``next()`` only makes the current stream wait for the copy.

Sampling: a permutation of the samples per epoch (numpy ``Generator(seed)``), continued across calls -- every sample is
seen once per epoch, where the reference's fresh shuffle per call samples with replacement across steps.
"""
import collections
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from ..data_preprocessing import SPEC_FRAMES, spec_dataset_preprocessing


def _load(dataset, i):
    try:
        return dataset[i]
    except FileNotFoundError:
        return None


class _EndlessBatches(torch.utils.data.Sampler):
    """Batch sampler of the streaming mode: an endless stream of index lists, recorded in hand-out order."""

    def __init__(self, owner):
        self.owner = owner

    def __iter__(self):
        o = self.owner
        while True:
            ids = [next(o._order) for _ in range(o.batch_size)]
            o._issued.append(ids)
            yield ids


class DeviceLoader:
    def __init__(self, dataset, batch_size, shuffle=True, device="cuda", resident=None, workers=8, prefetch=3, seed=0,
                 budget_bytes=64 << 30, collate_fn=spec_dataset_preprocessing):
        self.dataset, self.batch_size, self.shuffle = dataset, int(batch_size), shuffle
        self.device = torch.device(device)
        self.collate_fn = collate_fn
        self.workers, self.prefetch = max(1, workers), max(1, prefetch)
        self._rng = np.random.default_rng(seed)
        self._pool = ThreadPoolExecutor(max_workers=self.workers)
        n = len(dataset)
        if n == 0:
            raise ValueError("DeviceLoader: empty dataset")
        if resident is None:
            first = dataset[0]
            per_sample = sum(t[:, :SPEC_FRAMES].numel() * t.element_size() for t in first[:3]) + first[5].numel() * 4
            resident = per_sample * n <= budget_bytes
        self.resident = bool(resident)
        self.last_indices = None
        if self.resident:
            self._load_resident()
        else:
            self._order = self._index_stream(n)
            self._issued = collections.deque()          # index lists in the order the batch sampler handed them out
            cuda = self.device.type == "cuda"
            w = int(workers)
            self._torch_loader = torch.utils.data.DataLoader(
                dataset, batch_sampler=_EndlessBatches(self), collate_fn=collate_fn, num_workers=w, pin_memory=cuda,
                persistent_workers=w > 0, prefetch_factor=4 if w > 0 else None)
            self._batches = iter(self._torch_loader)
            self._ready = queue.Queue(maxsize=self.prefetch)
            self._copy_stream = torch.cuda.Stream(self.device) if cuda else None
            self._stop = False
            self._feeder = threading.Thread(target=self._feed, daemon=True)
            self._feeder.start()

    # ------------------------------------------------------------------------------------------------ sampling
    def _index_stream(self, n):
        """Endless stream of sample indices: one permutation (or 0..n-1) per epoch."""
        while True:
            for i in (self._rng.permutation(n) if self.shuffle else np.arange(n)):
                yield int(i)

    # ------------------------------------------------------------------------------------------------ resident
    def _load_resident(self):
        n = len(self.dataset)
        items = list(self._pool.map(lambda i: _load(self.dataset, i), range(n)))
        keep = [(i, it) for i, it in enumerate(items) if it is not None and it[0].shape[1] >= SPEC_FRAMES]
        if not keep:
            raise ValueError("DeviceLoader: no sample has %d frames" % SPEC_FRAMES)
        self.sample_ids = [i for i, _ in keep]
        cols = list(zip(*[it for _, it in keep]))
        stack = lambda specs: torch.stack([s[:, :SPEC_FRAMES] for s in specs]).to(self.device)   # noqa: E731
        self._speech, self._rir, self._echoed = stack(cols[0]), stack(cols[1]), stack(cols[2])
        self._fs = torch.stack([torch.as_tensor(f) for f in cols[3]]).to(self.device)
        self._theta = torch.stack(cols[4]).to(self.device)
        self._wiener = torch.stack(cols[5]).to(self.device)
        self._order = self._index_stream(len(keep))

    def _next_resident(self):
        pos = [next(self._order) for _ in range(self.batch_size)]
        self.last_indices = [self.sample_ids[p] for p in pos]
        idx = torch.as_tensor(pos, device=self.device)
        return tuple(t.index_select(0, idx) for t in (self._speech, self._rir, self._echoed, self._fs, self._theta,
                                                        self._wiener))

    # ------------------------------------------------------------------------------------------------ streaming
    def _feed(self):
        while not self._stop:
            batch = next(self._batches)               # collated (and pinned) by a worker process, in sampler order
            ids = self._issued.popleft()
            if isinstance(batch[0], list):            # every sample of the batch was too short: take the next one
                continue
            event = None
            if self._copy_stream is not None:
                with torch.cuda.stream(self._copy_stream):
                    batch = tuple(t.to(self.device, non_blocking=True) for t in batch)
                    event = torch.cuda.Event()
                    event.record(self._copy_stream)
            else:
                batch = tuple(batch)
            while not self._stop:
                try:
                    self._ready.put((ids, batch, event), timeout=0.2)
                    break
                except queue.Full:
                    pass

    def _next_streaming(self):
        ids, batch, event = self._ready.get()
        self.last_indices = ids
        if event is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(event)
            for t in batch:
                t.record_stream(cur)
        return batch

    # ------------------------------------------------------------------------------------------------ iterator
    def __iter__(self):
        return self

    def __next__(self):
        return self._next_resident() if self.resident else self._next_streaming()

    def __len__(self):
        n = len(self.sample_ids) if self.resident else len(self.dataset)
        return max(1, n // self.batch_size)

    def close(self):
        if not self.resident:
            self._stop = True
            try:
                while True:
                    self._ready.get_nowait()
            except queue.Empty:
                pass
            self._feeder.join(timeout=2.0)
            self._batches = None                      # releases the worker processes
            self._torch_loader = None
        self._pool.shutdown(wait=False)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
