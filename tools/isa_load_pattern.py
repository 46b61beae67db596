"""Order of vector-memory loads, waits, stores, barriers and branches in every kernel of a gfx950 assembly file (hipcc -save-temps):
a quick look at whether a kernel's loads go out together ("LLLLw3w2w1w0") or one by one ("Lw0.Lw0.Lw0"): the second shape pays
one memory round trip per load.  usage: tools/isa_load_pattern.py file.s [...]"""
import re
import sys

for f in sys.argv[1:]:
    txt = open(f).read()
    for m in re.finditer(r'^(_ZN2cd\w+):[^\n]*\n(.*?)s_endpgm', txt, re.S | re.M):
        name = re.sub(r'_ZN2cd\d+', '', m.group(1))
        seq = []
        for l in m.group(2).split('\n'):
            l = l.strip()
            if l.startswith(('global_load', 'buffer_load')):
                seq.append('L')
            elif l.startswith('s_waitcnt') and 'vmcnt' in l:
                seq.append('w' + re.search(r'vmcnt\((\d+)\)', l).group(1))
            elif l.startswith('s_barrier'):
                seq.append('|')
            elif l.startswith('global_store'):
                seq.append('S')
            elif l.startswith('s_cbranch'):
                seq.append('.')
        s = re.sub(r'\.+', '.', ''.join(seq))
        print(name[:26].ljust(26), s[:260])
