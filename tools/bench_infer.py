"""BASELINE configs[4]-shaped inference run (1 GPU): zero-shot evaluation set with variable lengths (log-uniform
32..2048) padded to bucket sizes, EC-sized label table (5134 labels x 2 descriptions, ensembled) swapped in at
run time after a GO-sized table, no model re-creation.  Prints a JSON line (not the driver's bench contract)."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from bench import build_model

BUCKETS = (128, 256, 512, 1024, 2048)


def main(n_seq=1024, batch=128):
    dev = torch.device("cuda", 0)
    model = build_model(dev).eval()
    model.inference_descriptions_per_label = 2
    g = torch.Generator().manual_seed(5)
    lens = torch.exp(torch.rand(n_seq, generator=g) * (math.log(2048) - math.log(32)) + math.log(32)).long().clamp(32, 2048)
    ids = torch.randint(0, 20, (n_seq, 2048), generator=g)
    tables = {"GO-2019 (32102 x 2)": torch.randn(32102 * 2, 1024, generator=g).to(dev),
              "EC (5134 x 2)": torch.randn(5134 * 2, 1024, generator=g).to(dev)}
    batches = []
    for bi, bmax in enumerate(BUCKETS):
        lo = BUCKETS[bi - 1] if bi else 0
        rows = torch.nonzero((lens > lo) & (lens <= bmax)).flatten()
        for s in range(0, len(rows), batch):
            r = rows[s:s + batch]
            x = torch.nn.functional.one_hot(ids[r, :bmax], 20).permute(0, 2, 1).float().contiguous()
            for k, i in enumerate(r):
                x[k, :, lens[i]:] = 0
            batches.append((x.to(dev), lens[r].to(dev)))
    out = {}
    with torch.no_grad():
        for name, table in tables.items():
            model(sequence_onehots=batches[0][0], sequence_lengths=batches[0][1], label_embeddings=table)  # warm-up
            torch.cuda.synchronize()
            t0 = time.time()
            for x, l in batches:
                logits, _ = model(sequence_onehots=x, sequence_lengths=l, label_embeddings=table)
            torch.cuda.synchronize()
            dt = time.time() - t0
            nl = table.shape[0] // 2
            out[name] = {"seconds": dt, "sequences_per_s": n_seq / dt, "pairs_per_s": n_seq * table.shape[0] / dt,
                         "logits_shape": [int(logits.shape[0]), int(nl)]}
    print(json.dumps({"workload": f"{n_seq} sequences, lengths log-uniform 32..2048 in buckets {BUCKETS}, batch {batch}",
                      "residues": int(lens.sum()), "results": out}))


if __name__ == "__main__":
    main()
