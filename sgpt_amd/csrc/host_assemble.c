/* Host leg f1 of SURVEY 8: the BEIR result dict {qid: {doc_id: score}} from the device's [nq, k] (score, corpus position)
 * lists -- custommodels/exact_search.py:109-132 builds it entry by entry in Python.  The reference driver asks for
 * k = 1000 (+1) (beir_dense_retriever.py:440): a million dict entries per 1000 queries, 137-165 ms of CPython per 3.3 ms of
 * device pass when done as `dict(zip(ids[idx].tolist(), vals.tolist()))` per row (round 4).  This is the same construction
 * in C: dicts pre-sized for k entries (no rehash on the way), no intermediate object array / key and value lists, and the
 * id strings -- a random walk over the corpus id list -- prefetched a few entries ahead of the insert that hashes them.
 * CPython extension (PyObjects in, PyObjects out), not part of the C ABI of include/sgpt_hip.h; sgpt_amd/beir.py falls back to
 * the Python construction when it is not built (same dict, same insertion order). */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>

/* Pre-sized dicts are a private CPython call: used where it is known to exist with this signature (CPython 3.8 - 3.12, the GIL
 * build); anywhere else -- newer, free-threaded or non-CPython interpreters -- the public constructor (the dict then grows by
 * rehashing: slower, same contents).  List items are read through the public macro. */
#if defined(Py_GIL_DISABLED) || !defined(PY_VERSION_HEX) || PY_VERSION_HEX < 0x03080000 || PY_VERSION_HEX >= 0x030D0000 || \
    defined(PYPY_VERSION) || defined(Py_LIMITED_API)
#define SGPT_DICT_NEW(n) PyDict_New()
#else
#define SGPT_DICT_NEW(n) _PyDict_NewPresized(n)
#endif

/* assemble(query_ids: list, corpus_ids: list, vals: C-contiguous float32[nq, k], idxs: C-contiguous int64[nq, k]) -> dict
 * Entries with idx < 0 are padding (a shard with fewer than k documents) and are skipped, exact_search.py:117-120. */
static PyObject* assemble(PyObject* self, PyObject* args) {
    PyObject *qids, *cids, *vobj, *iobj;
    (void)self;
    if (!PyArg_ParseTuple(args, "O!O!OO", &PyList_Type, &qids, &PyList_Type, &cids, &vobj, &iobj)) return NULL;
    Py_buffer vb, ib;
    if (PyObject_GetBuffer(vobj, &vb, PyBUF_C_CONTIGUOUS) < 0) return NULL;
    if (PyObject_GetBuffer(iobj, &ib, PyBUF_C_CONTIGUOUS) < 0) { PyBuffer_Release(&vb); return NULL; }
    PyObject* out = NULL;
    const Py_ssize_t nq = PyList_GET_SIZE(qids), nc = PyList_GET_SIZE(cids);
    if (vb.itemsize != 4 || ib.itemsize != 8 || nq == 0 || vb.len % (4 * nq) || vb.len / 4 != ib.len / 8) {
        PyErr_SetString(PyExc_ValueError, "assemble: vals float32[nq, k] and idxs int64[nq, k] for len(query_ids) = nq expected");
        goto done;
    }
    {
        const Py_ssize_t k = vb.len / 4 / nq;
        const float* v = (const float*)vb.buf;
        const int64_t* ix = (const int64_t*)ib.buf;
        out = SGPT_DICT_NEW(nq);
        if (!out) goto done;
        for (Py_ssize_t q = 0; q < nq; ++q) {
            PyObject* d = SGPT_DICT_NEW(k);
            if (!d) { Py_CLEAR(out); goto done; }
            const float* vr = v + q * k;
            const int64_t* ir = ix + q * k;
            for (Py_ssize_t j = 0; j < k; ++j) {
                if (j + 16 < k) {
                    const int64_t pn = ir[j + 16];
                    if (pn >= 0 && pn < nc) __builtin_prefetch(PyList_GET_ITEM(cids, pn));
                }
                const int64_t p = ir[j];
                if (p < 0) continue;
                if (p >= nc) {
                    PyErr_SetString(PyExc_IndexError, "assemble: corpus position out of range");
                    Py_DECREF(d); Py_CLEAR(out); goto done;
                }
                PyObject* f = PyFloat_FromDouble((double)vr[j]);
                if (!f || PyDict_SetItem(d, PyList_GET_ITEM(cids, p), f) < 0) { Py_XDECREF(f); Py_DECREF(d); Py_CLEAR(out); goto done; }
                Py_DECREF(f);
            }
            if (PyDict_SetItem(out, PyList_GET_ITEM(qids, q), d) < 0) { Py_DECREF(d); Py_CLEAR(out); goto done; }
            Py_DECREF(d);
        }
    }
done:
    PyBuffer_Release(&vb);
    PyBuffer_Release(&ib);
    return out;
}

static PyMethodDef methods[] = {{"assemble", assemble, METH_VARARGS, "{qid: {doc_id: score}} from [nq, k] score / position arrays"},
                                {NULL, NULL, 0, NULL}};
static struct PyModuleDef mod = {PyModuleDef_HEAD_INIT, "_sgpt_host", NULL, -1, methods, NULL, NULL, NULL, NULL};
PyMODINIT_FUNC PyInit__sgpt_host(void) { return PyModule_Create(&mod); }
