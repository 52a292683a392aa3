"""Extracts (name, type, default) of every command-line flag of the reference into flags.json.
Run in the build container (the reference is not available on the GPU box):

    python tests/golden/make_flags.py /root/reference/main.py tests/golden/flags.json
"""
import json
import re
import sys


def main(src, dst):
    text = open(src).read()
    rows = re.findall(r"add_argument\('--(\w+)', type=(\w+), default=([^,]+?)(?:,| ,)", text)
    out = [[n, t, eval(d.strip())] for n, t, d in rows]
    json.dump(out, open(dst, "w"), indent=0)
    print(len(out), "flags")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
