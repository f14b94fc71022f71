"""Instruction mix of the largest loop of a kernel in a hipcc -S listing.   python tools/isa_loop_mix.py file.s <mangled-name-substring> ..."""
import collections
import sys


def main():
    s = open(sys.argv[1]).read().split('\n')
    for name in sys.argv[2:]:
        starts = [i for i, l in enumerate(s) if name in l.split(':')[0] and l.startswith('_Z') and ':' in l]
        if not starts:
            print(name, "not found")
            continue
        start = starts[0]
        end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
        lines = [l.split(';')[0].strip() for l in s[start + 1:end]]
        lines = [l for l in lines if l and not (l.startswith('.') and not l.endswith(':'))]
        labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(':')}
        best = (0, 0, 0)
        for i, l in enumerate(lines):
            if l.startswith('s_cbranch') or l.startswith('s_branch'):
                tgt = l.split()[-1]
                if tgt in labels and labels[tgt] < i and i - labels[tgt] > best[0]:
                    best = (i - labels[tgt], labels[tgt], i)
        loop = lines[best[1]:best[2]]
        c = collections.Counter()
        for l in loop:
            if l.endswith(':'):
                continue
            op = l.split()[0]
            for pre, key in (('v_mfma', 'mfma'), ('scratch_load', 'scratch_load'), ('scratch_store', 'scratch_store'),
                             ('v_cvt_pk_bf16', 'cvt_pk'), ('v_exp', 'exp'), ('v_accvgpr', 'accvgpr'), ('v_', 'valu'), ('ds_', 'ds'),
                             ('global_', 'vmem'), ('buffer_', 'vmem'), ('s_waitcnt', 'waitcnt'), ('s_nop', 'nop'), ('s_', 'salu')):
                if op.startswith(pre):
                    c[key] += 1
                    break
        print(name, 'loop instructions', len(loop), dict(c))


if __name__ == "__main__":
    main()
