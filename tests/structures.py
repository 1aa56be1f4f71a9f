"""Jacobians of awkward shapes shared by the CPU and GPU parity tests (test infrastructure)."""
import numpy as np
import scipy.sparse as sp

# Iteration counts at the reference's sqrt(eps) tolerances: the first four stop at the same iteration whatever the rounding
# (measured: tests/test_oracle.py, the restatement in three summation orders; the device); the last three run into an
# episode of lost orthogonality right where a stopping test fires (profiles/r04_fixed_iteration_probe.txt) and may stop an
# iteration or two apart under a different -- equally valid -- rounding.
WELL_CONDITIONED = ["tiny", "square-ish", "wide-window", "duplicates-free-unsorted"]
ORDER_SENSITIVE = ["dense-row", "dense-column", "empty-columns"]
ALL_KINDS = WELL_CONDITIONED + ORDER_SENSITIVE


def random_structure(kind, rng):
    """Jacobians of awkward shapes (full row rank with probability one): what the storage layouts have to cope with."""
    if kind == "tiny":              # fewer entries than one lane group
        m, n = 1, 3
        A = sp.csr_matrix(rng.standard_normal((m, n)))
    elif kind == "square-ish":      # m close to n, short rows of A'
        m, n = 180, 200
        A = sp.random(m, n, density=0.04, random_state=np.random.RandomState(3), format="csr") + sp.eye(m, n) * 3.0
    elif kind == "wide-window":     # A' blocks span more than 8192 columns: 16-bit columns, but not the column-sorted layout
        m, n = 10000, 12000
        rows = np.repeat(np.arange(m), 3)
        cols = np.concatenate([np.arange(m), (np.arange(m) * 7919) % n, (np.arange(m) * 104729 + 13) % n]).reshape(3, m).T.ravel()
        A = sp.csr_matrix((rng.standard_normal(3 * m) + np.tile([4.0, 0.0, 0.0], m), (rows, cols)), shape=(m, n))
    elif kind == "empty-columns":   # columns of A without entries = empty rows of A' (empty row blocks)
        m, n = 300, 5000
        A = sp.random(m, 600, density=0.03, random_state=np.random.RandomState(5), format="csr") + sp.eye(m, 600) * 2.0
        A = sp.hstack([A, sp.csr_matrix((m, n - 600))], format="csr")
    elif kind == "dense-row":       # one constraint touching every variable: a long row of A, a 1-entry-heavier A'
        m, n = 120, 6000
        A = sp.vstack([sp.random(m - 1, n, density=0.004, random_state=np.random.RandomState(7), format="csr")
                       + sp.eye(m - 1, n) * 2.0, sp.csr_matrix(np.ones((1, n)))], format="csr")
    elif kind == "dense-column":    # one variable in every constraint: a row of A' longer than an LDS stage
        m, n = 3000, 9000
        A = sp.random(m, n, density=0.0015, random_state=np.random.RandomState(9), format="lil")
        A[:, 17] = rng.standard_normal((m, 1))
        A = sp.csr_matrix(A) + sp.eye(m, n) * 2.0
    elif kind == "duplicates-free-unsorted":  # CSR with unsorted column indices inside the rows
        m, n = 400, 3000
        A = sp.random(m, n, density=0.01, random_state=np.random.RandomState(11), format="csr") + sp.eye(m, n) * 2.0
        A = sp.csr_matrix(A)
        for i in range(m):
            a, b = A.indptr[i], A.indptr[i + 1]
            perm = rng.permutation(b - a)
            A.indices[a:b] = A.indices[a:b][perm]
            A.data[a:b] = A.data[a:b][perm]
    else:
        raise ValueError(kind)
    A = sp.csr_matrix(A)
    A.has_sorted_indices = kind != "duplicates-free-unsorted"
    return A
