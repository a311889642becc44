#!/usr/bin/env python3
"""Resolve the ablation / what-if conditionals of the kernel sources at their shipped values (a small unifdef).

Rounds 2-4 measured alternatives inside the product kernels behind -D switches (V2W_WS_PERSIST, V2W_BF_ABL_NOFRAG, V2W_CT_EXP_NOMFMA ...);
the measurements are in DESIGN.md, the switches are dead weight in what ships.  This tool rewrites a source with every conditional that
tests ONLY such macros resolved (the taken branch stays, the others go); `#ifndef X / #define X v / #endif` default blocks collapse to the
plain `#define` (tuning constants stay named).  Conditionals on anything else (V2W_TIMELINE: the diagnostic stamp build) are left alone.

    python tools/strip_ablations.py wavthruvec_pytorch_amd/csrc/*.hip        # in place
"""
import re
import sys

UNDEFINED = {'V2W_BF_ABL_NOCOMMIT', 'V2W_BF_ABL_NOFRAG', 'V2W_BF_ABL_NOEPI', 'V2W_RS_BB2', 'V2W_RS_ABL_NOEPI', 'V2W_WS_ABL_NOA', 'V2W_WS_ABL_NOB',
             'V2W_WS_NOCT', 'V2W_WS_C16', 'V2W_NO_N16', 'V2W_TL_NOGLOAD', 'V2W_TL_NODSREAD', 'V2W_TL_NOLRELU', 'V2W_CT_EXP_NOMFMA',
             'V2W_CT_EXP_NOEPI', 'V2W_SPLIT_NAB', 'V2W_SPLIT_FORCE'}
# macros with a default value defined in the file (`#ifndef X / #define X v / #endif`): conditionals on them resolve at that value
VALUED = {'V2W_WS_PERSIST', 'V2W_WS_PRIO', 'V2W_WS_PRE', 'V2W_WS_CFG', 'V2W_WS_RING', 'V2W_RS_C64_CFG', 'V2W_SPLIT_C64', 'V2W_SPLIT_WPE',
          'V2W_PM_WGS', 'V2W_PM_NT', 'V2W_PM_DEPTH', 'V2W_N16_WN'}
KNOWN = UNDEFINED | VALUED
DIRECTIVE = re.compile(r'^\s*#\s*(if|ifdef|ifndef|elif|else|endif)\b(.*)$')


def evaluate(kind, expr, values):
    """True / False when `expr` tests known macros only, else None."""
    expr = re.sub(r'//.*$', '', expr).strip()
    names = set(re.findall(r'\b[A-Za-z_]\w*\b', expr)) - {'defined'}
    if not names or not names <= KNOWN:
        return None
    if kind == 'ifdef':
        return expr in values
    if kind == 'ifndef':
        return expr not in values
    e = re.sub(r'defined\s*\(\s*(\w+)\s*\)', lambda m: str(int(m.group(1) in values)), expr)
    e = re.sub(r'\b[A-Za-z_]\w*\b', lambda m: str(values.get(m.group(0), 0)), e)
    e = e.replace('&&', ' and ').replace('||', ' or ').replace('!', ' not ').replace(' not =', '!=')
    return bool(eval(e))


def strip(text):
    values = {}
    out = []
    stack = []          # per open conditional: dict(resolved, taken, emitting, parent_emit)
    emitting = True
    for line in text.split('\n'):
        m = DIRECTIVE.match(line)
        if not m:
            if emitting:
                d = re.match(r'^\s*#\s*define\s+(\w+)\s+(-?\d+)\b', line)
                if d and d.group(1) in VALUED:
                    values[d.group(1)] = int(d.group(2))
                out.append(line)
            continue
        kind, expr = m.group(1), m.group(2)
        if kind in ('if', 'ifdef', 'ifndef'):
            r = evaluate(kind, expr, values) if emitting else None
            stack.append(dict(resolved=r is not None and emitting, taken=bool(r), parent=emitting))
            if stack[-1]['resolved']:
                emitting = bool(r)
            elif emitting:
                out.append(line)
        elif kind == 'elif':
            top = stack[-1]
            if top['resolved']:
                if top['taken']:
                    emitting = False
                else:
                    r = evaluate('if', expr, values)
                    if r is None:
                        raise SystemExit(f'cannot resolve #elif {expr}')
                    top['taken'] = bool(r)
                    emitting = top['parent'] and bool(r)
            elif top['parent']:
                out.append(line)
        elif kind == 'else':
            top = stack[-1]
            if top['resolved']:
                emitting = top['parent'] and not top['taken']
                top['taken'] = True
            elif top['parent']:
                out.append(line)
        else:
            top = stack.pop()
            if top['resolved']:
                emitting = top['parent']
            elif top['parent']:
                out.append(line)
    assert not stack
    return '\n'.join(out)


if __name__ == '__main__':
    for path in sys.argv[1:]:
        src = open(path).read()
        new = strip(src)
        if new != src:
            open(path, 'w').write(new)
            print(f'{path}: {src.count(chr(10)) - new.count(chr(10))} lines removed')
