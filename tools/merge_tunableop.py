"""merge_tunableop.py NEW.csv: the result lines of a TunableOp run into settlers_of_catan_rl_amd/tunableop_gfx950.csv (same validator
header required; a line for a shape that is already there replaces it)."""
import os, sys
base = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "settlers_of_catan_rl_amd", "tunableop_gfx950.csv")
b = open(base).read().splitlines(); n = open(sys.argv[1]).read().splitlines()
assert b[:5] == n[:5], (b[:5], n[:5])
key = lambda l: ",".join(l.split(",")[:2])
d = {key(l): l for l in b[5:]}; order = [key(l) for l in b[5:]]; added = 0
for l in n[5:]:
    k = key(l)
    if k not in d:
        order.append(k); added += 1
    d[k] = l
open(base, "w").write("\n".join(b[:5] + [d[k] for k in order]) + "\n")
print("lines", len(order), "added", added)
