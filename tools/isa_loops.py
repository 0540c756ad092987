"""Loops of one kernel in a hipcc -S listing: instruction mix of every backward-branch region.
usage: python tools/isa_loops.py listing.s mangled-name-substring [min-instructions]"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
names = [n for n in re.findall(r'\n(_Z\w+):', s) if pat in n]
for nm in names:
    i = s.index('\n' + nm + ':'); j = s.index('.Lfunc_end', i)
    lines = s[i:j].split('\n')
    lab = {}
    for n, l in enumerate(lines):
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m: lab[m.group(1)] = n
    def mix(a, b):
        ins = [l.strip() for l in lines[a:b + 1] if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
        v = [l for l in ins if l.startswith('v_')]
        mov = [l for l in v if l.startswith(('v_mov_b32', 'v_mov_b64', 'v_accvgpr', 'v_cndmask'))]
        f64 = [l for l in v if '_f64' in l]
        return len(ins), len(v), len(f64), len(mov), len([l for l in ins if l.startswith('s_')]), len([l for l in ins if l.startswith('ds_')]), len([l for l in ins if l.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))])
    print(nm, 'whole kernel: %d instrs, %d valu (%d f64, %d mov/cndmask), %d salu, %d ds, %d vmem' % mix(0, len(lines) - 1))
    for n, l in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in lab and lab[m.group(1)] < n:
            a = lab[m.group(1)]
            r = mix(a, n)
            if r[0] >= mn: print('  loop %5d-%5d: %d instrs, %d valu (%d f64, %d mov/cndmask), %d salu, %d ds, %d vmem' % ((a, n) + r))
