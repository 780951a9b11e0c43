#!/usr/bin/env python3
"""oracle/patch_material.py -- TEST INFRASTRUCTURE.  Builds the reference-on-GPU oracle for "the reference's program with a
user's closest-hit and miss shaders": writes, into a temporary directory, the reference's samples/shader.cl with the BODY of
its `material` function replaced by tests/golden/user_material.inc (and, when a third argument is given, the body of
`environment` by that file), and prints that directory.  oracle/Makefile compiles ref_gpu_dev.cl against it (-I<dir> before
-I$(REF)/samples) into oracle/_ref/ref_shader_gfx950_um.co and removes the directory: the patched text never enters the
repository, only the compiled object travels (like the other _ref/ files).
"""
import os, re, sys, tempfile


def replace_body(text, fn, body):
    m = re.search(r"\bvoid\s+" + fn + r"\s*\(", text)
    assert m, "no `%s` definition" % fn
    i = text.index("{", text.index(")", m.end()))
    depth, j = 0, i
    while True:
        if text[j] == "{":
            depth += 1
        elif text[j] == "}":
            depth -= 1
            if depth == 0:
                break
        j += 1
    return text[:i] + body + text[j + 1:]


ref = sys.argv[1]
text = open(os.path.join(ref, "samples", "shader.cl")).read()
text = replace_body(text, "material", open(sys.argv[2]).read())
if len(sys.argv) > 3:
    text = replace_body(text, "environment", open(sys.argv[3]).read())
out = tempfile.mkdtemp(prefix="rdx_um_")
open(os.path.join(out, "shader.cl"), "w").write(text)
print(out)
