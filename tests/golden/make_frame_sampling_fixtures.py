"""Golden vectors for the reference's frame-sampling helpers (reference utils/utils.py:201-229: uniform_sample, get_sparse_indices,
get_dense_indices -- what evaluation/*/inference_*.py use to pick the frames handed to the MLLM and to SAM2).  utils/utils.py cannot be
imported in this container (it pulls torchvision / matplotlib at module level), so the three function definitions are taken from the
reference file AT GENERATION TIME with ast and executed against numpy; nothing of the reference's text is stored here or in the fixture.
    python tests/golden/make_frame_sampling_fixtures.py"""
import ast
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open("/root/reference/utils/utils.py").read()
mod = ast.parse(src)
want = {"uniform_sample", "get_sparse_indices", "get_dense_indices"}
fns = [n for n in mod.body if isinstance(n, ast.FunctionDef) and n.name in want]
assert {f.name for f in fns} == want
ns = {"np": np}
exec(compile(ast.Module(body=fns, type_ignores=[]), "reference:utils/utils.py", "exec"), ns)

rows_u, rows_s, rows_d = [], [], []
for total in list(range(1, 70)) + [100, 128, 257, 1000, 1801]:
    for n in (1, 2, 3, 4, 8, 15, 16, 31, 32, 64):
        if n <= total:
            rows_u.append([total, n] + ns["uniform_sample"](total, n) + [-1] * (64 - n))
        rows_s.append([total, n] + ns["get_sparse_indices"](total, n) + [-1] * (64 - n))
for nm in (4, 8, 15, 16, 32, 64):
    for nsam in range(1, min(nm, 33)):
        rows_d.append([nm, nsam] + ns["get_dense_indices"](nm, nsam) + [-1] * (32 - nsam))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frame_sampling.npz"), uniform=np.asarray(rows_u, np.int64), sparse=np.asarray(rows_s, np.int64),
                    dense=np.asarray(rows_d, np.int64))
print(len(rows_u), len(rows_s), len(rows_d))
