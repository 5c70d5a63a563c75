"""Static instruction mix of the kernels in an ISA listing: python tools/isa_mix.py file.s [name-substring]"""
import re, sys, collections
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else ''
cur = None; stats = {}; meta = {}
for l in lines:
    m = re.match(r'^(_Z\w+):', l)
    if m:
        cur = m.group(1); stats[cur] = collections.Counter(); continue
    if cur is None: continue
    t = l.strip()
    m = re.match(r'; (NumVgprs|NumAgprs|NumSgprs|ScratchSize|Occupancy|TotalNumVgprs|LDSByteSize|codeLenInByte): (\d+)', t)
    if m: meta.setdefault(cur, {})[m.group(1)] = int(m.group(2))
    if not t or t.startswith(';') or t.startswith('.'): continue
    op = t.split()[0]
    if re.match(r'^[sv]_|^ds_|^global_|^buffer_|^flat_|^scratch_', op): stats[cur][op] += 1
def cat(op):
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('flat_') or op.startswith('buffer_') or op.startswith('scratch_'): return 'vmem'
    if re.match(r'v_(fma|fmac|mul|add|sub|mad|pk_fma|pk_mul|pk_add)_f(32|64)', op): return 'fp'
    return 'valu_other'
for k, c in stats.items():
    tot = sum(c.values())
    if tot < 200 or pat not in k: continue
    cc = collections.Counter()
    for op, n in c.items(): cc[cat(op)] += n
    print(k[:110], tot, meta.get(k, {}))
    print('   ', dict(cc))
    print('   ', c.most_common(14))
