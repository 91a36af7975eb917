"""Instruction mix of one kernel in a hipcc -S listing: tools/asm_counts.py file.s <mangled-name-fragment> ..."""
import re
import sys

s = open(sys.argv[1]).read()
for key in sys.argv[2:]:
    m = re.search(r'^(_ZN5iqhip\w*' + re.escape(key) + r'\w*):[^\n]*\n', s, re.M)
    if not m:
        print("not found:", key)
        continue
    body = s[m.end():s.index('s_endpgm', m.end())]
    c = lambda pat: len(re.findall(pat, body))  # noqa: E731
    print(m.group(1)[:90])
    print("   st_x4 %d st_x2 %d st_other %d | ld_x4 %d ld_x2 %d ld_other %d | dpp %d cndmask %d | mfma16 %d mfma4 %d | v_f64 %d | "
          "ds_read %d ds_write %d | s_waitcnt %d vmcnt0 %d | scratch %d | lines %d" % (
              c(r'global_store_dwordx4'), c(r'global_store_dwordx2'), c(r'global_store_(?!dwordx[24])'),
              c(r'global_load_dwordx4'), c(r'global_load_dwordx2'), c(r'global_load_(?!dwordx[24])'),
              c(r'quad_perm'), c(r'v_cndmask'), c(r'v_mfma_f64_16x16x4'), c(r'v_mfma_f64_4x4x4'),
              c(r'\bv_(mul|fma|add|max|min)_f64'), c(r'ds_read'), c(r'ds_write'), c(r's_waitcnt'), c(r'vmcnt\(0\)'),
              c(r'scratch_'), body.count('\n')))
