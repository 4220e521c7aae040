"""Identity of the kernel sources a counter file was collected from: sha256 over csrc/*.hip, csrc/*.h and include/gsplat.h
(sorted by name).  profiles/summarize.py stores it in pmc_traffic.json / sq_insts.json; bench.py recomputes it and refuses
to combine counters of OTHER sources with a live duration (git HEAD does not travel to the GPU box, and a docs-only commit
should not invalidate a profile)."""
import glob
import hashlib
import os


def source_id(root=None):
    root = root or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "sparse-view-3dgs-pack_amd", "csrc")
    files = sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")) +
                   [os.path.join(root, "include", "gsplat.h")])
    h = hashlib.sha256()
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_id())
