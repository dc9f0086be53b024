"""Dump the hottest loop (the basic block with the most MFMAs) of one kernel from hipcc -S output.
usage: isa_loop.py file.s <substring of mangled kernel name> [--full]"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
# split in basic blocks
blocks, cur = [], []
for l in body:
    if re.match(r"^\.LBB\S+:", l) and cur:
        blocks.append(cur)
        cur = []
    cur.append(l)
    if re.match(r"^\s+s_(cbranch|branch|endpgm)", l):
        blocks.append(cur)
        cur = []
if cur:
    blocks.append(cur)
best = max(blocks, key=lambda b: sum("v_mfma" in l for l in b))
VALU = re.compile(r"\s+v_")
SALU = re.compile(r"\s+s_")
print(f"# kernel lines {len(body)}, blocks {len(blocks)}, hottest block: {len(best)} lines, "
      f"{sum('v_mfma' in l for l in best)} mfma, {sum('global_load_lds' in l for l in best)} dma, "
      f"{sum('ds_read' in l for l in best)} ds_read, {sum(bool(VALU.match(l)) and 'mfma' not in l for l in best)} valu, "
      f"{sum(bool(SALU.match(l)) for l in best)} salu")
for l in body:
    m = re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l)
for l in lines[end:end + 60]:
    m = re.search(r"\.(num_vgpr|num_agpr|numbered_sgpr|private_seg_size), (\d+)", l)
    if m:
        print("#", m.group(1), m.group(2))
if "--full" in sys.argv:
    print("\n".join(l for l in best if not l.strip().startswith(";")))
