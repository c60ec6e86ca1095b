"""TEST INFRASTRUCTURE.  Rewrites the few ISPC-only constructs of kernel.ispc into C++ at build time:

    python translate.py <kernel.ispc> <out.cpp>      (out lives under oracle/_ref/, git-ignored; never committed)

Rules (everything else passes through byte for byte, line numbers kept):
  1. `foreach (i = a ... b)`  ->  `for (int i = a; i < (b); i++)`     one program instance walks the range
  2. float literals get an `f` suffix (`0.5` -> `0.5f`, `1f` -> `1.0f`, `1e99` -> `1e99f`): ISPC literals are fp32
  3. a function that is defined twice, differing only in uniform / varying qualifiers (an ISPC overload pair that
     collapses to one C++ signature), keeps its first definition
  4. `T name[0] = { ... }` -> `T name[] = { ... }`: ISPC represents an unsized array by element count 0, so the
     `float rgb_span[0] = { 0, 0, 0 }` of kernel.ispc:3057 IS the unsized declaration, sized by its initialiser
The qualifiers, `export` and the sized int names are macros in ispc_prelude.h."""
import re
import sys

src = open(sys.argv[1], encoding="latin-1").read().split("\n")

FOREACH = re.compile(r"foreach\s*\(\s*(\w+)\s*=\s*(.+?)\s*\.\.\.\s*(.+)\)\s*$")
LITERAL = re.compile(r"(?<![\w.])((?:\d+\.\d*|\.\d+)(?:[eE][+-]?\d+)?|\d+[eE][+-]?\d+)(?![\w.])")
INT_F = re.compile(r"(?<![\w.])(\d+)f\b")


def fix_code(code):
    m = FOREACH.search(code)
    if m:
        code = code[:m.start()] + "for (int %s = %s; %s < (%s); %s++)" % (m.group(1), m.group(2), m.group(1), m.group(3), m.group(1))
    code = re.sub(r"\[0\](\s*=\s*\{)", r"[]\1", code)          # rule 4
    code = INT_F.sub(r"\1.0f", code)
    code = LITERAL.sub(r"\1f", code)
    return code


out = []
in_block_comment = False
for line in src:
    # split off comments so that rule 2 never touches them
    code, rest = line, ""
    if in_block_comment:
        end = line.find("*/")
        if end < 0:
            out.append(line)
            continue
        rest_head, code = line[:end + 2], line[end + 2:]
        in_block_comment = False
        prefix = rest_head
    else:
        prefix = ""
    cut = len(code)
    for tok in ("//", "/*"):
        i = code.find(tok)
        if 0 <= i < cut:
            cut = i
    code, rest = code[:cut], code[cut:]
    if rest.startswith("/*") and "*/" not in rest:
        in_block_comment = True
    out.append(prefix + fix_code(code) + rest)

# rule 3: duplicate definitions after qualifier erasure
QUAL = re.compile(r"\b(uniform|varying|const)\b")
text = out
seen = {}
i = 0
result = []
while i < len(text):
    line = text[i]
    m = re.match(r"^(inline\s+)?[\w ]+?\b(\w+)\s*\((.*)\)\s*$", line)
    if m and i + 1 < len(text) and text[i + 1].strip() == "{" and not line.startswith((" ", "\t")):
        key = (m.group(2), re.sub(r"\s+", " ", QUAL.sub("", m.group(3))).strip().replace("* ", "*").replace(" *", "*"))
        depth, j = 0, i + 1
        while True:
            depth += text[j].count("{") - text[j].count("}")
            j += 1
            if depth == 0:
                break
        if key in seen:
            result.extend("// (ISPC overload of line %d, identical once the qualifiers are erased)" % seen[key] for _ in range(i, j))
            i = j
            continue
        seen[key] = i + 1
    result.append(line)
    i += 1

open(sys.argv[2], "w").write('#line 1 "%s"\n' % sys.argv[1] + "\n".join(result))
