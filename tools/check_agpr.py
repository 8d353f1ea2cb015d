#!/usr/bin/env python3
"""check_agpr.py <file.s> <kernel name substring>: fails when the compiler-generated part of the
kernel (everything outside ;;#ASMSTART ... ;;#ASMEND) names an accumulation register, or when the kernel
uses scratch memory (a spill).  stream4_kernel (mlp_stream4.hip) keeps live data in AGPRs between its asm statements."""
import re
import sys


def main(path, kern):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % kern, l)]
    if not starts:
        sys.exit("kernel %s not found in %s" % (kern, path))
    failed = False
    for st in starts:
        en = next(i for i in range(st, len(lines)) if ".amdhsa_kernel" in lines[i] and kern in lines[i])
        inside, bad, n = False, [], 0
        for l in lines[st:en]:
            t = l.strip()
            if "ASMSTART" in t:
                inside = True
                continue
            if "ASMEND" in t:
                inside = False
                continue
            if not t or t.startswith((";", ".", "//")):
                continue
            n += 1
            if not inside and re.search(r"\ba\d+\b|\ba\[\d+", t.split(";")[0]):
                bad.append(t)
        info = "\n".join(lines[en:en + 250])
        get = lambda k: int(re.search(r"; %s: (\d+)" % k, info).group(1))   # noqa: E731
        print("%s: %d instructions, %d VGPRs + %d AGPRs, occupancy %d, scratch %d, AGPR uses outside asm: %d"
              % (lines[st].split(":")[0][-48:], n, get("NumVgprs"), get("NumAgprs"), get("Occupancy"), get("ScratchSize"), len(bad)))
        if bad:
            print("\n".join(bad[:20]))
            failed = True
        if get("ScratchSize") > 0:
            print("kernel spills: ScratchSize %d" % get("ScratchSize"))
            failed = True
    if failed:
        sys.exit(1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
