"""Compact instruction sequence of one kernel's ISA: python scratch/isa_seq.py file.s mangled_name_substring
M mfma, e exp, R ds_read_b128, T ds_read_b64_tr, r other ds_read, W ds_write, w s_waitcnt, B s_barrier, D buffer_load, c cvt, m v_mul_f32,
P s_setprio, v other VALU, J branch, . other"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
on = False
out = []
for ln in txt:
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        on = sub in m.group(1)
        continue
    if not on: continue
    if '.end_amdhsa_kernel' in ln: break
    t = ln.strip()
    if not t or t.startswith(';') or t.startswith('.') and not t.startswith('.LBB'): continue
    op = t.split()[0]
    if op.startswith('.LBB'):
        out.append('\n' + op + ' ' + ('<loop>' if 'Loop Header' in ln else '') + '\n'); continue
    if 'v_mfma' in op: c = 'M'
    elif 'v_exp' in op: c = 'e'
    elif 'ds_read_b128' in op: c = 'R'
    elif 'ds_read_b64_tr' in op: c = 'T'
    elif op.startswith('ds_read'): c = 'r'
    elif op.startswith('ds_write'): c = 'W'
    elif op == 's_waitcnt': c = 'w'
    elif op == 's_barrier': c = 'B'
    elif op.startswith('buffer_load'): c = 'D'
    elif op.startswith('v_cvt'): c = 'c'
    elif op.startswith('v_mul_f32'): c = 'm'
    elif op == 's_setprio': c = 'P'
    elif op.startswith('v_'): c = 'v'
    elif op.startswith('s_cbranch') or op == 's_branch': c = 'J'
    else: c = '.'
    out.append(c)
print(''.join(out))
