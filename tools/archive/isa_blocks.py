"""Basic-block census of one kernel in a device assembly file (hipcc --cuda-device-only -S): registers, and for every block of at
least MIN instructions its VALU / scalar / LDS / memory instruction counts and a few tell-tale opcodes.
usage: python tools/isa_blocks.py <file.s> <kernel name substring> [min instructions = 25]"""
import re
import sys

s = open(sys.argv[1]).read()
name = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 25
m = re.search(r'^(_Z\w*' + re.escape(name) + r'\w*):[^\n]*\n', s, re.M)
sym = m.group(1)
start = m.end()
end = s.index('s_endpgm', start)
body = s[start:end].split('\n')
meta = re.search(r'\.amdhsa_kernel ' + re.escape(sym) + r'.*?\.end_amdhsa_kernel', s, re.S).group(0)
print(sym[:80], len(body), 'lines')
for l in meta.split('\n'):
    if any(x in l for x in ('next_free_vgpr', 'next_free_sgpr', 'accum_offset', 'private_segment_fixed_size', 'group_segment_fixed')):
        print('  ', l.strip())
blocks = []
cur = ['entry', []]
blocks.append(cur)
for l in body:
    if re.match(r'^\.LBB\d+_\d+:', l):
        cur = [l.strip(), []]
        blocks.append(cur)
    elif l.strip() and not l.strip().startswith(';') and not l.strip().startswith('.'):
        cur[1].append(l.strip())
tot = {}
for nm, ins in blocks:
    kinds = {}
    for i in ins:
        op = i.split()[0]
        key = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'lds' if op.startswith('ds_') else 'mem'
        kinds[key] = kinds.get(key, 0) + 1
        tot[key] = tot.get(key, 0) + 1
    if len(ins) >= mn:
        tags = {t: sum(t in i for i in ins) for t in ('ds_add', 'v_rcp', 'v_sqrt', 'v_pk_', 'v_cvt', 'v_cmp_lt_u64', 'ds_read_b128', 'ds_read', 'ds_write', 'v_sad', 'v_cndmask', 'v_fma', 'v_cmp', 'scratch_')}
        print(nm, len(ins), kinds, {k: v for k, v in tags.items() if v})
print('total', tot)
