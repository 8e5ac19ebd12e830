/* fastobj.c — the result objects of the batched request path, built in one C loop (CPython C API; module rag_project_icd10_amd._fastobj).
 *
 * MultiDiagnosisService.match_diagnoses_batch (row N2 of SURVEY.md section 8) turns the winners of 1 000 diagnosis strings into
 * 10 000 Candidate objects with 10 000 SimilarityFactors - what the reference builds one by one from rescored hit dicts
 * (services/multi_diagnosis_service.py:161-175, models/icd_models.py:56-87). The values arrive as Python floats / ints already
 * (tolist() of the device results); creating the objects from Python costs ~1.3 us each, most of what that path still costs the
 * host. This is the same construction - the object layout api/icd_models.py trusted_candidate checks against the validated
 * constructor once per process - without the interpreter between the fields. Optional: the package falls back to the Python
 * loop (api/icd_models.py bulk_candidates) when this module is not built.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>

static PyObject *k_code, *k_title, *k_score, *k_level, *k_parent, *k_enh, *k_orig, *k_fact;
static PyObject *k_vs, *k_hb, *k_em, *k_sc, *k_ca, *k_cr;
static PyObject *k_dict, *k_fields_set, *k_extra, *k_private;
static PyObject *v_one, *v_empty, *v_zero;

/* an instance of a plain heap type (a dataclass: object.__new__ is tp_alloc) whose __dict__ is `d` (reference stolen) */
static PyObject *instance_with_dict(PyTypeObject *tp, PyObject *d) {
    PyObject *o = tp->tp_alloc(tp, 0);
    if (!o) { Py_DECREF(d); return NULL; }
    PyObject **dp = _PyObject_GetDictPtr(o);
    if (!dp) { Py_DECREF(d); Py_DECREF(o); PyErr_SetString(PyExc_TypeError, "type has no __dict__"); return NULL; }
    Py_XSETREF(*dp, d);
    return o;
}

/* bulk_candidates(candidate_type, factors_type, fields_set, codes, titles, ids, scores, originals, vs, hb, sc, cr) -> list
 * ids / scores / originals / vs / hb: lists (the first len(ids) entries of the others are used). Raises ValueError on a score
 * that is negative or NaN (the caller lets the validated constructor raise the reference's ValidationError). */
static PyObject *bulk_candidates(PyObject *self, PyObject *args) {
    PyObject *ctype, *ftype, *fields, *codes, *titles, *ids, *scores, *origs, *vs, *hb, *sc, *cr;
    if (!PyArg_ParseTuple(args, "OOOO!O!O!O!O!O!O!OO", &ctype, &ftype, &fields, &PyList_Type, &codes, &PyList_Type, &titles, &PyList_Type, &ids,
                          &PyList_Type, &scores, &PyList_Type, &origs, &PyList_Type, &vs, &PyList_Type, &hb, &sc, &cr))
        return NULL;
    if (!PyType_Check(ctype) || !PyType_Check(ftype)) { PyErr_SetString(PyExc_TypeError, "types expected"); return NULL; }
    const Py_ssize_t n = PyList_GET_SIZE(ids), nrows = PyList_GET_SIZE(codes);
    if (PyList_GET_SIZE(scores) < n || PyList_GET_SIZE(origs) < n || PyList_GET_SIZE(vs) < n || PyList_GET_SIZE(hb) < n || PyList_GET_SIZE(titles) != nrows) {
        PyErr_SetString(PyExc_ValueError, "parallel lists are shorter than ids");
        return NULL;
    }
    PyObject *out = PyList_New(n);
    if (!out) return NULL;
    for (Py_ssize_t j = 0; j < n; ++j) {
        PyObject *s = PyList_GET_ITEM(scores, j);
        const double sv = PyFloat_AsDouble(s);
        if (sv == -1.0 && PyErr_Occurred()) goto fail;
        if (!(sv >= 0.0)) { PyErr_SetString(PyExc_ValueError, "negative or NaN score"); goto fail; }
        const Py_ssize_t row = PyLong_AsSsize_t(PyList_GET_ITEM(ids, j));
        if (row < 0 || row >= nrows) { if (!PyErr_Occurred()) PyErr_SetString(PyExc_IndexError, "row id outside the corpus"); goto fail; }
        PyObject *fd = _PyDict_NewPresized(6);
        if (!fd) goto fail;
        if (PyDict_SetItem(fd, k_vs, PyList_GET_ITEM(vs, j)) || PyDict_SetItem(fd, k_hb, PyList_GET_ITEM(hb, j)) || PyDict_SetItem(fd, k_em, v_zero) ||
            PyDict_SetItem(fd, k_sc, sc) || PyDict_SetItem(fd, k_ca, v_zero) || PyDict_SetItem(fd, k_cr, cr)) { Py_DECREF(fd); goto fail; }
        PyObject *f = instance_with_dict((PyTypeObject *)ftype, fd);
        if (!f) goto fail;
        PyObject *cd = _PyDict_NewPresized(8);
        if (!cd) { Py_DECREF(f); goto fail; }
        if (PyDict_SetItem(cd, k_code, PyList_GET_ITEM(codes, row)) || PyDict_SetItem(cd, k_title, PyList_GET_ITEM(titles, row)) || PyDict_SetItem(cd, k_score, s) ||
            PyDict_SetItem(cd, k_level, v_one) || PyDict_SetItem(cd, k_parent, v_empty) || PyDict_SetItem(cd, k_enh, s) ||
            PyDict_SetItem(cd, k_orig, PyList_GET_ITEM(origs, j)) || PyDict_SetItem(cd, k_fact, f)) { Py_DECREF(f); Py_DECREF(cd); goto fail; }
        Py_DECREF(f);
        PyTypeObject *ct = (PyTypeObject *)ctype;
        PyObject *c = ct->tp_alloc(ct, 0);
        if (!c) { Py_DECREF(cd); goto fail; }
        /* the four slots of a pydantic v2 model, set the way object.__setattr__ sets them (not through BaseModel.__setattr__) */
        if (PyObject_GenericSetAttr(c, k_dict, cd) || PyObject_GenericSetAttr(c, k_fields_set, fields) ||
            PyObject_GenericSetAttr(c, k_extra, Py_None) || PyObject_GenericSetAttr(c, k_private, Py_None)) { Py_DECREF(cd); Py_DECREF(c); goto fail; }
        Py_DECREF(cd);
        PyList_SET_ITEM(out, j, c);
    }
    return out;
fail:
    Py_DECREF(out);
    return NULL;
}

static PyMethodDef methods[] = {
    {"bulk_candidates", bulk_candidates, METH_VARARGS, "the Candidate objects (with their SimilarityFactors) of one query's winners"},
    {NULL, NULL, 0, NULL}};
static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_fastobj", "bulk construction of the request path's result objects", -1, methods};

PyMODINIT_FUNC PyInit__fastobj(void) {
#define K(var, text) if (!(var = PyUnicode_InternFromString(text))) return NULL
    K(k_code, "code"); K(k_title, "title"); K(k_score, "score"); K(k_level, "level"); K(k_parent, "parent_code"); K(k_enh, "enhanced_score");
    K(k_orig, "original_score"); K(k_fact, "similarity_factors");
    K(k_vs, "vector_similarity"); K(k_hb, "hierarchy_boost"); K(k_em, "entity_match_score"); K(k_sc, "semantic_coherence");
    K(k_ca, "category_alignment"); K(k_cr, "context_relevance");
    K(k_dict, "__dict__"); K(k_fields_set, "__pydantic_fields_set__"); K(k_extra, "__pydantic_extra__"); K(k_private, "__pydantic_private__");
#undef K
    v_one = PyLong_FromLong(1); v_empty = PyUnicode_InternFromString(""); v_zero = PyFloat_FromDouble(0.0);
    if (!v_one || !v_empty || !v_zero) return NULL;
    return PyModule_Create(&moduledef);
}
