"""Per kernel: MFMA count, AGPR<->VGPR copies, register split.  python scratch/isa_acc_report.py file.s ..."""
import re, sys
for f in sys.argv[1:]:
    lines = open(f).read().split('\n')
    name, cnt = None, None
    for ln in lines:
        m = re.match(r'^(_Z\w+):\s', ln)
        if m:
            name, cnt = m.group(1), dict(m=0, r=0, w=0)
            continue
        if name is None:
            continue
        if 'v_mfma' in ln: cnt['m'] += 1
        elif 'v_accvgpr_read' in ln: cnt['r'] += 1
        elif 'v_accvgpr_write' in ln: cnt['w'] += 1
        mm = re.match(r'; (NumVgprs|NumAgprs|Occupancy|ScratchSize): (\d+)', ln)
        if mm:
            cnt[mm.group(1)] = int(mm.group(2))
            if mm.group(1) == 'Occupancy':
                if cnt['m']:
                    print(f"{f.split('/')[-1][:-2]:16s} {name[:78]:78s} mfma {cnt['m']:4d} accR {cnt['r']:4d} accW {cnt['w']:4d} V {cnt.get('NumVgprs')} A {cnt.get('NumAgprs')} scr {cnt.get('ScratchSize')} occ {cnt['Occupancy']}")
                name = None
