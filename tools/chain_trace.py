"""Per-item timeline of the chained launches (csrc/conv_chain.hip, experiment build, YOLO_CHAIN_TRACE=<file>).
python tools/chain_trace.py <file>: per launch -- span, per layer kind: tiles, mean wait for dependencies, mean run time; slot utilisation."""
import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
pos = 0; launch = 0
HDR = 4 + 24 + 1
while pos < raw.size:
    groups, n_items, n_layers = int(raw[pos]), int(raw[pos + 1]), int(raw[pos + 2])
    first = raw[pos + 4: pos + 4 + n_layers + 1].astype(np.int64)
    pos += HDR
    rec = raw[pos: pos + groups * n_items * 4].reshape(groups, n_items, 4).astype(np.int64)
    pos += groups * n_items * 4
    launch += 1
    if len(sys.argv) > 2 and launch <= int(sys.argv[2]): continue      # skip the first launches (warm-up)
    t0 = rec[:, :, 0].min()
    claim, ready, end = [(rec[:, :, k] - t0) / 100.0 for k in range(3)]
    span = end.max()
    busy = (end - claim).sum()
    print("launch %d: %d layers, %d items/group, span %.1f us, slot-time %.0f us = %.2f of 512 slots x span; waiting %.0f us (%.1f %%)" % (
        launch, n_layers, n_items, span, busy, busy / (512 * span), (ready - claim).sum(), 100 * (ready - claim).sum() / busy))
    for j in range(n_layers):
        sl = slice(first[j], first[j + 1])
        w = (ready[:, sl] - claim[:, sl]); r = (end[:, sl] - ready[:, sl])
        print("  layer %2d: %4d tiles/group  wait mean %6.2f max %6.1f   run mean %6.2f p10 %6.2f p90 %6.2f   first claim %7.1f last end %7.1f" % (
            j, first[j + 1] - first[j], w.mean(), w.max(), r.mean(), np.percentile(r, 10), np.percentile(r, 90), claim[:, sl].min(), end[:, sl].max()))
    if len(sys.argv) > 3:       # detail of one layer: start-time deciles and the run time of the tiles starting in each
        j = int(sys.argv[3]); sl = slice(first[j], first[j + 1])
        st = ready[:, sl].ravel(); r = (end[:, sl] - ready[:, sl]).ravel()
        order = np.argsort(st)
        for q in range(10):
            idx = order[q * len(order) // 10:(q + 1) * len(order) // 10]
            print("    layer %d decile %d: start %7.1f .. %7.1f  run mean %6.2f min %6.2f max %6.2f" % (j, q, st[idx].min(), st[idx].max(), r[idx].mean(), r[idx].min(), r[idx].max()))
    ge = end.max(axis=1)
    print("  per-group end:", np.array2string(ge, precision=1))
