"""A small tokenized `train_folder` in InternEvo's on-disk format, generated deterministically (never committed):

    <root>/cn/a.bin  <root>/cn/a.bin.meta        one JSON document per line {"tokens": [...]}; the .meta file is np.save of an
    <root>/en/b.bin  <root>/en/b.bin.meta        int array [n_docs, 2] = (byte offset of the line, number of tokens)
    <root>/en/c.bin  <root>/en/c.bin.meta        (tools/tokenizer.py of the reference writes exactly this pair)

Used by make_golden.py --data-folder (the real reference pipeline reads it) and by tests/test_oracle_golden.py (this repo's reader)."""
import json
import os

import numpy as np


def write_folder(root, seed=7):
    rng = np.random.RandomState(seed)
    spec = {"cn/a.bin": 180, "en/b.bin": 150, "en/c.bin": 90}
    for rel, n_docs in spec.items():
        path = os.path.join(root, rel)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        meta, cur = [], 0
        with open(path, "wb") as f:
            for _ in range(n_docs):
                n = int(rng.choice([3, 9, 40, 77, 130, 300], p=[0.1, 0.2, 0.3, 0.2, 0.15, 0.05])) + int(rng.randint(0, 7))
                toks = rng.randint(1, 500, size=n).tolist()
                if rng.rand() < 0.1:
                    toks[int(rng.randint(0, n))] *= -1  # negative ids mark "no loss here" in the reference's data (collaters.py)
                line = (json.dumps({"tokens": toks}) + "\n").encode()
                f.write(line)
                meta.append((cur, n))
                cur += len(line)
        with open(path + ".meta", "wb") as f:
            np.save(f, np.array(meta, dtype=np.int64))
    return spec
