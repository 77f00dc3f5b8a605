"""Host-side mirror of the reference's synthetic-data path for the packed hot path.

Same observable batches as InternEvo's
  RandomDataset            internlm/data/tokenized/dummy_dataset.py:8-49
  PackedDatasetWithCut     internlm/data/tokenized/packed_dataset.py:204-331 (build_pack)
  StaticBatchSampler       internlm/data/tokenized/batch_sampler.py:110-247
  JsonlDataset             internlm/data/tokenized/single_dataset.py:18-117   (`data.train_folder`: tokenized .bin + .meta files)
  get_packed_dataset_without_short_length   internlm/data/tokenized/packed_dataset.py:393-480
  validation loaders       internlm/data/build_dataloader.py:67-157 + DataParallelSampler (batch_sampler.py:20-93) + jsonl_ds_collate_fn
  packed_collate_fn        internlm/data/tokenized/collaters.py:7-58
but built for a 288 GB-HBM node with modest host RAM: samples are generated on demand from the
pre-drawn (n, r) arrays instead of materialising a million Python lists (the reference needs
13-21 GB RSS and 2-5 minutes per rank at seq 4096), and a batch is assembled
straight into int64 numpy buffers.  tests/test_oracle_golden.py pins the first batches against fixtures
produced by the real reference pipeline (tests/golden/data.json).
"""
import bisect
import itertools
import json
import mmap
import os
import re

import numpy as np
import torch

DEFAULT_SEED = 1024  # internlm/data/tokenized/packed_dataset.py:21


class RandomDataset:
    """dummy_dataset.py:8-49.  tokens(i) = ([n, r] + list(range(n)) * r)[:max_len] with r doubled until the
    sample is at least max_len long when fixed_seqlen."""

    def __init__(self, num_samples=10000, max_len=1024, fixed_seqlen=False):
        rng = np.random.RandomState(1999)
        self.max_num = rng.randint(1, 30, size=(num_samples,))
        self.rep_num = rng.randint(10, 200, size=(num_samples,))
        self.max_len = max_len
        n, r = self.max_num.astype(np.int64), self.rep_num.astype(np.int64)
        if fixed_seqlen:
            # `while len(d) < max_len: r *= 2` -- the check is on n*r BEFORE the [n, r] prefix is added
            r = r.copy()
            need = n * r < max_len
            while need.any():
                r[need] *= 2
                need = n * r < max_len
        self._r = r
        self.lengths = np.minimum(n * r + 2, max_len).astype(int)

    def __len__(self):
        return len(self.lengths)

    def tokens(self, index, start=0, stop=None):
        """tokens[start:stop] of sample `index` as an int64 array (no materialisation of the full list)."""
        n, r = int(self.max_num[index]), int(self._r[index])
        length = int(self.lengths[index])
        stop = length if stop is None else min(stop, length)
        pos = np.arange(start, stop, dtype=np.int64)
        out = (pos - 2) % n
        if start < 2:
            head = np.array([n, r], dtype=np.int64)[start : min(2, stop)]
            out[: len(head)] = head
        return out

    def token_at(self, index, pos):
        n, r = int(self.max_num[index]), int(self._r[index])
        return n if pos == 0 else (r if pos == 1 else (pos - 2) % n)


class JsonlDataset:
    """single_dataset.py:18-117: one .bin file = one JSON document per line ({"tokens": [...]}), `<file>.meta` = np.save of
    [n_docs, k] with the byte offset of each line first and its token count last; documents shorter than min_length are dropped.
    Same access surface as RandomDataset (lengths / tokens / token_at) so PackedDatasetWithCut packs either."""

    def __init__(self, path, type_id=0, min_length=50):
        self.path = os.path.realpath(path)
        meta_path = self.path + ".meta"
        assert os.path.exists(meta_path), f"The cache file:{meta_path} is not found for file:{path}"
        with open(meta_path, "rb") as f:
            meta = np.load(f)
        self.offsets, self.lengths = meta[:, 0], meta[:, -1]
        self.type_id = type_id
        self.old_length = len(self.offsets)
        if min_length > 0:
            keep = self.lengths >= min_length
            self.offsets, self.lengths = self.offsets[keep], self.lengths[keep]
        self._mm, self._last = None, (-1, None)

    def __len__(self):
        return len(self.offsets)

    def _doc(self, index):
        if self._last[0] != index:
            if self._mm is None:
                with open(self.path, "rb") as f:
                    self._mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            self._mm.seek(int(self.offsets[index]))
            line = self._mm.readline().decode("utf-8")
            try:
                toks = json.loads(line)["tokens"]
            except Exception as err:  # noqa: BLE001
                raise ValueError(f"Error while loading JSONL line in file {self.path} at byte {int(self.offsets[index])}: {err}") from err
            self._last = (index, np.asarray(toks, dtype=np.int64))
        return self._last[1]

    def tokens(self, index, start=0, stop=None):
        return self._doc(index)[start:stop]

    def token_at(self, index, pos):
        return int(self._doc(index)[pos])


class PackedDatasetWithCut:
    """packed_dataset.py:204-331, packed (`use_packed_dataset=True`) path only."""

    def __init__(self, dataset, max_length_per_sample=2048, packed_length=4096):
        self.dataset = dataset
        self.max_length_per_sample = max_length_per_sample
        self.packed_length = packed_length
        self.lengths = dataset.lengths
        rng = np.random.RandomState(DEFAULT_SEED)
        self.sample_indices = np.arange(len(self.lengths))
        rng.shuffle(self.sample_indices)
        self.len_samples_shuffled = self.lengths[self.sample_indices]
        self.acm_len_samples = np.cumsum(self.len_samples_shuffled)
        self.num_tokens = int(self.lengths.sum())

    def __len__(self):
        return self.num_tokens // self.packed_length

    def _cal_map(self, carriage_idx):
        return int(np.searchsorted(self.acm_len_samples, (carriage_idx + 1) * self.packed_length, side="left"))

    def _mapping(self, pack_idx):
        pre_pos, pre_token_id = 0, 0
        if pack_idx > 0:
            pre_pos = self._cal_map(pack_idx - 1)
            pre_token_id = int(self.len_samples_shuffled[pre_pos] - (self.acm_len_samples[pre_pos] - pack_idx * self.packed_length))
            if pre_token_id == self.len_samples_shuffled[pre_pos]:
                pre_pos += 1
                pre_token_id = 0
        pos = self._cal_map(pack_idx)
        token_id = int(self.len_samples_shuffled[pos] - (self.acm_len_samples[pos] - (pack_idx + 1) * self.packed_length))
        return pre_pos, pre_token_id, pos, token_id

    def __getitem__(self, item):
        pre_pos, pre_token_id, pos, token_id = self._mapping(item)
        toks, labs, cu, idxs = [], [], [0], []
        mlen = self.max_length_per_sample

        def close_chunk(n):
            full, left = divmod(n, mlen)
            for _ in range(full):
                cu.append(cu[-1] + mlen)
                idxs.append(np.arange(mlen, dtype=np.int64))
            if left > 0:
                cu.append(cu[-1] + left)
                idxs.append(np.arange(left, dtype=np.int64))

        while pre_pos < pos:
            si = int(self.sample_indices[pre_pos])
            chunk = self.dataset.tokens(si, pre_token_id)
            toks.append(chunk)
            labs.append(np.concatenate([chunk[1:], np.array([-100], dtype=np.int64)]))
            close_chunk(len(chunk))
            pre_pos += 1
            pre_token_id = 0
        si = int(self.sample_indices[pos])
        chunk = self.dataset.tokens(si, pre_token_id, token_id)
        toks.append(chunk)
        last = -100 if token_id == int(self.dataset.lengths[si]) else self.dataset.token_at(si, token_id)
        labs.append(np.concatenate([chunk[1:], np.array([last], dtype=np.int64)]))
        close_chunk(len(chunk))
        return {
            "tokens": np.concatenate(toks),
            "labels": np.concatenate(labs),
            "cu_seqlens": np.array(cu, dtype=np.int32),
            "indexes": np.concatenate(idxs) if idxs else np.zeros(0, dtype=np.int64),
            "type_ids": np.full(self.packed_length, getattr(self.dataset, "type_id", 0), dtype=np.int64),  # one type per file
        }


class PackedDatasetWithoutCuSeqlen:
    """packed_dataset.py:70-202 (`data.pack_sample_into_one = True`): the shuffled documents are laid end to end and cut into packs
    of packed_length tokens; a pack is presented as packed_length / seq_len sequences of exactly seq_len tokens (cu_seqlens at
    multiples of seq_len, positions restarting with them) whatever the document boundaries are; a document piece's last label
    is -100 even when the document continues in the next pack."""

    def __init__(self, dataset, max_length_per_sample=2048, packed_length=4096):
        assert packed_length % max_length_per_sample == 0
        self.dataset = dataset
        self.max_length_per_sample, self.packed_length = max_length_per_sample, packed_length
        self.bsz = packed_length // max_length_per_sample
        self.lengths = dataset.lengths
        rng = np.random.RandomState(DEFAULT_SEED)
        self.indices = np.arange(len(self.lengths))
        rng.shuffle(self.indices)
        self.cum_lens = np.cumsum(self.lengths[self.indices])
        self.num_tokens = int(self.lengths.sum())

    def __len__(self):
        return self.num_tokens // self.packed_length

    def _find_offset(self, offset):
        idx = int(np.searchsorted(self.cum_lens, offset, side="right"))
        return (idx, offset) if idx == 0 else (idx, int(offset - self.cum_lens[idx - 1]))

    def __getitem__(self, item):
        s_idx, s_len = self._find_offset(item * self.packed_length)
        e_idx, e_len = self._find_offset((item + 1) * self.packed_length)
        pieces = []
        if s_idx == e_idx:
            pieces.append(self.dataset.tokens(int(self.indices[s_idx]), s_len, e_len))
        else:
            pieces.append(self.dataset.tokens(int(self.indices[s_idx]), s_len))
            pieces += [self.dataset.tokens(int(self.indices[i])) for i in range(s_idx + 1, e_idx)]
            if e_len:
                pieces.append(self.dataset.tokens(int(self.indices[e_idx]), 0, e_len))
        S = self.max_length_per_sample
        return {
            "tokens": np.concatenate(pieces),
            "labels": np.concatenate([np.concatenate([p[1:], np.array([-100], dtype=np.int64)]) for p in pieces]),
            "cu_seqlens": np.arange(self.bsz + 1, dtype=np.int32) * S,
            "indexes": np.tile(np.arange(S, dtype=np.int64), self.bsz),
            "type_ids": np.full(self.packed_length, getattr(self.dataset, "type_id", 0), dtype=np.int64),
        }


class ConcatPacked:
    """torch.utils.data.ConcatDataset over the per-file packed datasets."""

    def __init__(self, datasets):
        self.datasets = list(datasets)
        self.cum = list(itertools.accumulate(len(d) for d in self.datasets))

    def __len__(self):
        return self.cum[-1] if self.cum else 0

    def __getitem__(self, i):
        k = bisect.bisect_right(self.cum, i)
        return self.datasets[k][i - (self.cum[k - 1] if k else 0)]


def dataset_type_ids_map(folder):
    """data/utils.py:11-14: the sub-folders of train_folder, sorted, are the dataset types ("en", "cn", ...) of the metric."""
    return {key: idx for idx, key in enumerate(sorted(os.listdir(folder)))}


def build_folder_dataset(folder, max_length_per_sample, packed_length, min_length=0, min_length_dict=None, pack_sample_into_one=False):
    """packed_dataset.py:393-480: every .bin under `folder`, walked top-down with sorted directory
    and file names, becomes a JsonlDataset -> PackedDatasetWithCut (PackedDatasetWithoutCuSeqlen with pack_sample_into_one); files left empty by the
    length filter are skipped."""
    assert os.path.exists(folder), f"{folder} does not exist."
    type_map = dataset_type_ids_map(folder)
    packed = []
    for root, dirs, files in os.walk(folder, followlinks=True):
        dirs.sort()
        for fn in sorted(files):
            if not fn.endswith(".bin"):
                continue
            fp = os.path.join(root, fn)
            ml = min_length
            if min_length_dict is not None:
                hits = [k for k in min_length_dict if k in fp]
                assert len(hits) < 2, f"The file name `{fp}` matched the following resample keys:{hits}"
                ml = min_length_dict[hits[0]] if hits else ml
            match = [idx for key, idx in type_map.items() if re.search(rf"/[z_]*{key}/", fp)]  # data/utils.py:17-25
            assert len(match) == 1, f"{fp}, match_idxes should be 1, but got {match} from {type_map}"
            ds = JsonlDataset(fp, match[0], min_length=ml)
            if len(ds) == 0:
                continue
            packed.append((PackedDatasetWithoutCuSeqlen if pack_sample_into_one else PackedDatasetWithCut)(ds, max_length_per_sample, packed_length))
    return ConcatPacked(packed)


class StaticBatchSampler:
    """batch_sampler.py:110-247 without batch-size ramp-up (rampup_batch_size="" in every config of the path), including its
    state_dict / load_state_dict (:249-272): the content of a checkpoint's `sampler.pt`."""

    def __init__(self, num_samples, batch_size, seed=1024, data_rank=0, data_world_size=1):
        self.num_samples = num_samples
        self.batch_size = batch_size
        self.seed, self.epoch = seed, 0
        self.rng = np.random.RandomState(seed)
        self.data_rank, self.data_world_size = data_rank, data_world_size
        self.batch_count = 0
        self._get_indices()

    def _get_indices(self):
        indices = np.arange(self.num_samples)
        self.rng_state = self.rng.get_state()  # the generator BEFORE this epoch's shuffle: enough to regenerate `indices`
        self.rng.shuffle(indices)
        n = self.num_samples // (self.batch_size * self.data_world_size) * self.batch_size * self.data_world_size
        self.indices = indices[:n]
        assert len(self.indices) >= self.batch_size, "The number of samples should be larger than batch_size"
        self.consumed = 0

    def __iter__(self):
        while True:
            mine = self.indices[self.data_rank :: self.data_world_size]
            while self.consumed < len(mine):
                batch = mine[self.consumed : self.consumed + self.batch_size]
                self.consumed += len(batch)
                self.batch_count += 1
                yield batch
            self._get_indices()

    def state_dict(self):
        return {"batch_size": self.batch_size, "raw_rampup_batch_size": "", "rng_state": self.rng_state, "epoch": self.epoch,
                "seed": self.seed, "data_world_size": self.data_world_size, "num_consumed_samples_in_epoch": self.consumed,
                "batch_count": self.batch_count, "indices": self.indices}

    def load_state_dict(self, states):
        for name, mine in (("data_world_size", self.data_world_size), ("raw_rampup_batch_size", ""), ("seed", self.seed)):
            assert states[name] == mine, (name, states[name], mine)  # should not change (batch_sampler.py:265-266)
        self.rng.set_state(states["rng_state"])
        self._get_indices()  # regenerate this epoch's order from the saved generator state
        self.epoch = states["epoch"]
        self.batch_count = states["batch_count"]
        self.consumed = states["num_consumed_samples_in_epoch"]


def packed_collate(items, packed_length):
    """collaters.py:7-58: tokens -> abs(), labels <= 0 -> -100 (sic: label 0 is ignored too)."""
    xs = np.stack([np.abs(b["tokens"]) for b in items])
    ys = np.stack([np.where(b["labels"] > 0, b["labels"], -100) for b in items])
    assert xs.shape[1] == packed_length and ys.shape[1] == packed_length
    return {
        "input_ids": torch.from_numpy(xs),
        "cu_seqlens": [torch.from_numpy(b["cu_seqlens"]) for b in items],
        "indexes": torch.from_numpy(np.stack([b["indexes"] for b in items])),
        "type_ids": torch.from_numpy(np.stack([b["type_ids"] for b in items])),
    }, torch.from_numpy(ys)


class FolderLoader:
    """build_dataloader.py:26-66 for data.train_folder = <tokenized folder>: the same sampler and collate over the packed files."""

    def __init__(self, folder, seq_len, micro_bsz, micro_num, min_length=0, min_length_dict=None, data_rank=0, data_world_size=1, seed=1024,
                 pack_sample_into_one=False):
        self.packed_length = seq_len * micro_bsz
        self.ds = build_folder_dataset(folder, seq_len, self.packed_length, min_length, min_length_dict, pack_sample_into_one)
        self.dataset_types = list(dataset_type_ids_map(folder).keys())
        self.sampler = StaticBatchSampler(len(self.ds), micro_num, seed, data_rank, data_world_size)

    def __iter__(self):
        for idx in self.sampler:
            yield packed_collate([self.ds[int(i)] for i in idx], self.packed_length)


class SyntheticLoader:
    """build_dataloader.py:26-66 + :84-116 for train_folder=None: yields (batch_dict, labels) with
    micro_num packed rows of micro_bsz*seq_len tokens each, this rank's shard of every global batch."""

    def __init__(self, seq_len, micro_bsz, micro_num, fixed_seqlen=False, num_samples=1_000_000, data_rank=0, data_world_size=1,
                 seed=1024):
        self.packed_length = seq_len * micro_bsz
        base = RandomDataset(num_samples=num_samples, max_len=seq_len, fixed_seqlen=fixed_seqlen)
        self.ds = PackedDatasetWithCut(base, max_length_per_sample=seq_len, packed_length=self.packed_length)
        self.sampler = StaticBatchSampler(len(self.ds), micro_num, seed, data_rank, data_world_size)

    def __iter__(self):
        for idx in self.sampler:
            yield packed_collate([self.ds[int(i)] for i in idx], self.packed_length)


def jsonl_collate(items, max_length_per_sample):
    """collaters.py:58-88 (validation): truncate to seq_len, tokens -> abs(), labels = shifted tokens with non-positive ids and the
    padding ignored (-100), both zero- / -100-padded to exactly seq_len."""
    xs = torch.zeros(len(items), max_length_per_sample, dtype=torch.int64)
    ys = torch.full((len(items), max_length_per_sample), -100, dtype=torch.int64)
    for r, toks in enumerate(items):
        t = torch.as_tensor(np.asarray(toks[:max_length_per_sample], dtype=np.int64))
        xs[r, : len(t)] = t.abs()
        lab = torch.where(t > 0, t, torch.full_like(t, -100))
        ys[r, : len(t) - 1] = lab[1:]
    return {"input_ids": xs}, ys


class _DocConcat:
    """ConcatDataset of JsonlDatasets as get_dataset_dict builds it (documents, not packs)."""

    def __init__(self, parts):
        self.parts = parts
        self.cum = list(itertools.accumulate(len(d) for d in parts))

    def __len__(self):
        return self.cum[-1] if self.cum else 0

    def tokens(self, i):
        k = bisect.bisect_right(self.cum, i)
        return self.parts[k].tokens(i - (self.cum[k - 1] if k else 0))


def valid_datasets(seq_len, fixed_seqlen, data_world_size, valid_folder=None):
    """build_dataloader.py:67-83: no valid_folder -> {"val": RandomDataset(500 samples per data-parallel rank)}; a folder -> one
    entry per directory that holds .bin files (tokenized/dataset.py:10-56: sorted walk, every .bin, min_length 50)."""
    if not valid_folder:
        return {"val": RandomDataset(num_samples=data_world_size * 500, max_len=seq_len, fixed_seqlen=fixed_seqlen)}
    assert os.path.exists(valid_folder), f"folder `{valid_folder}` not exists"
    out = {}
    for root, dirs, files in os.walk(valid_folder, followlinks=True):
        dirs.sort()
        parts = [JsonlDataset(os.path.join(root, fn)) for fn in sorted(files) if fn.endswith(".bin")]
        if parts:
            out[os.path.basename(root)] = _DocConcat(parts)
    return out


class ValidLoader:
    """One validation set as build_valid_loader_with_data_type iterates it: DataParallelSampler(shuffle=False, drop_last=True)
    -> this rank's documents rank, rank + world, ...; DataLoader(batch_size, drop_last=True); jsonl_ds_collate_fn.
    batch_size = min(valid_micro_num * micro_bsz, len // world) rounded down to whole micro-batches; 0 = the set is skipped."""

    def __init__(self, dataset, seq_len, micro_bsz, valid_micro_num, data_rank=0, data_world_size=1):
        self.ds, self.seq_len = dataset, seq_len
        n, w = len(dataset), data_world_size
        self.batch_size = min(valid_micro_num * micro_bsz, n // w) // micro_bsz * micro_bsz
        if w > 1:  # the sampler only exists under data parallelism (get_dpsampler_dataloader)
            per_rank = -((n - w) // -w) if n % w else n // w  # drop_last: ceil((n - w) / w)
            self.indices = list(range(n))[: per_rank * w][data_rank : per_rank * w : w]
        else:
            self.indices = list(range(n))

    def __len__(self):
        return len(self.indices) // self.batch_size if self.batch_size else 0

    def __iter__(self):
        bs = self.batch_size
        for b in range(len(self)):
            yield jsonl_collate([self.ds.tokens(i) for i in self.indices[b * bs : (b + 1) * bs]], self.seq_len)


class BatchSkipper:
    """data.skip_batches (utils/common.py:165-190, used at train.py:187,208-212): "a-b,c" names the batch counts whose batch is drawn from the loader -- the sampler
    and the consumed-sample count move on -- but not trained on.  Intervals are inclusive and must come in ascending order (the reference asserts it)."""

    def __init__(self, skip_batches=""):
        spans = []
        for interval in (str(skip_batches).split(",") if skip_batches else ()):
            if "-" in interval:
                start, end = map(int, interval.split("-"))
            else:
                start = end = int(interval)
            if spans and spans[-1] > start:
                raise AssertionError(f"data.skip_batches = {skip_batches!r}: the intervals must be in ascending order")
            spans.extend((start, end + 1))
        self.spans = spans

    def __call__(self, batch_count):
        import bisect

        return bisect.bisect_right(self.spans, batch_count) % 2 == 1

