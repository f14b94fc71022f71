"""Instruction mix between the phase marks of a kernel (listing made with -DELG_MARKS: every ELG_STAMP is an assembler comment).
    hipcc ... -DELG_MARKS -S --cuda-device-only elg_fwd.hip -o fwd.s;  python tools/isa_phase_mix.py fwd.s <mangled-name-substring>
Static counts: an inner loop's body counts once (its trip count is printed next to the mark it sits behind)."""
import collections, sys


def kind(op):
    for pre, key in (('v_mfma', 'mfma'), ('scratch_', 'scratch'), ('v_exp', 'trans'), ('v_rcp', 'trans'), ('v_log', 'trans'), ('v_sqrt', 'trans'),
                     ('v_rsq', 'trans'), ('v_readlane', 'readlane'), ('v_readfirstlane', 'readlane'), ('v_writelane', 'writelane'), ('v_', 'valu'),
                     ('ds_', 'ds'), ('global_', 'vmem'), ('buffer_', 'vmem'), ('flat_', 'vmem'), ('s_waitcnt', 'waitcnt'), ('s_nop', 'nop'),
                     ('s_cbranch', 'branch'), ('s_branch', 'branch'), ('s_barrier', 'barrier'), ('s_', 'salu')):
        if op.startswith(pre):
            return key
    return 'other'


def main():
    s = open(sys.argv[1]).read().split('\n')
    name = sys.argv[2]
    start = next(i for i, l in enumerate(s) if name in l.split(':')[0] and l.startswith('_Z') and ':' in l)
    end = next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
    cur, counts, order = 'entry', collections.defaultdict(collections.Counter), ['entry']
    for l in s[start + 1:end]:
        if 'ELG_PHASE_MARK' in l:
            cur = 'after mark ' + l.split('ELG_PHASE_MARK')[1].strip()
            if cur not in order:
                order.append(cur)
            continue
        t = l.split(';')[0].strip()
        if not t or t.endswith(':') or t.startswith('.'):
            continue
        counts[cur][kind(t.split()[0])] += 1
    for k in order:
        c = counts[k]
        print(f"{k:16s} total {sum(c.values()):5d}  " + "  ".join(f"{a} {b}" for a, b in sorted(c.items(), key=lambda x: -x[1])))


if __name__ == "__main__":
    main()
