/* Host-side helper of the ASER plugins: the class-balanced draw of ClassBalancedRandomSampling.sample
 * (reference: utils/buffer/buffer_utils.py:86-118) as one C call.
 *
 * What the reference's loop observably depends on, and how it is kept:
 *   - dict order of class_index_cache            -> PyDict_Next (insertion order)
 *   - `slots - excluded` building a NEW set whose iteration order decides which slot a permutation index means
 *                                                -> the same CPython operation (PyNumber_Subtract) and CPython's own set iterator
 *   - one torch.randperm(len(eligible)) per non-empty class on torch's global CPU generator
 *                                                -> the generator's algorithm restated here (mt19937 word stream, randperm's swap
 *                                                   loop) on the state bytes of torch.get_rng_state(); the caller writes the
 *                                                   advanced state back with torch.set_rng_state().  `randperm_check` lets the
 *                                                   Python side compare this restatement with the installed torch before it is
 *                                                   trusted (plugins/buffer_utils.py falls back to the Python loop otherwise).
 * Only bookkeeping: no image data passes through here. */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

/* layout of at::CPUGeneratorImpl's serialized state (ATen/CPUGeneratorImpl.cpp, CPUGeneratorImplStateLegacy + 2 trailing fields) */
#define MT_N 624
#define MT_M 397
#define TORCH_STATE_BYTES 5056
typedef struct {
    uint64_t seed;
    int32_t left;
    int32_t seeded;
    uint64_t next;
    uint64_t state[MT_N];
} MtState;

static void mt_reload(MtState* s) {
    uint64_t* st = s->state;
    int k;
#define TWIST(u, v) (((((uint32_t)(u)) & 0x80000000u) | (((uint32_t)(v)) & 0x7fffffffu)) >> 1 ^ ((((uint32_t)(v)) & 1u) ? 0x9908b0dfu : 0u))
    for (k = 0; k < MT_N - MT_M; ++k) st[k] = (uint32_t)st[k + MT_M] ^ TWIST(st[k], st[k + 1]);
    for (; k < MT_N - 1; ++k) st[k] = (uint32_t)st[k + MT_M - MT_N] ^ TWIST(st[k], st[k + 1]);
    st[MT_N - 1] = (uint32_t)st[MT_M - 1] ^ TWIST(st[MT_N - 1], st[0]);
#undef TWIST
    s->left = MT_N;
    s->next = 0;
}

static inline uint32_t mt_next(MtState* s) {
    if (--s->left == 0) mt_reload(s);
    uint32_t y = (uint32_t)s->state[s->next++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* torch.randperm(n) on the CPU generator (aten/src/ATen/native/TensorFactories.cpp, randperm_cpu, n < 2^32 / 20) */
static void randperm(MtState* s, int64_t n, int64_t* r) {
    for (int64_t i = 0; i < n; ++i) r[i] = i;
    for (int64_t i = 0; i < n - 1; ++i) {
        const int64_t z = (int64_t)(mt_next(s) % (uint32_t)(n - i));   /* n < 2^32 / 20: 32-bit arithmetic is exact */
        const int64_t t = r[i];
        r[i] = r[z + i];
        r[z + i] = t;
    }
}

static int state_from_buffer(Py_buffer* b, MtState** out) {
    if (b->len != TORCH_STATE_BYTES || b->readonly) {
        PyErr_SetString(PyExc_ValueError, "rng state must be a writable buffer of 5056 bytes (torch.get_rng_state())");
        return -1;
    }
    MtState* s = (MtState*)b->buf;
    if (s->left < 1 || s->left > MT_N || s->next > MT_N) {
        PyErr_SetString(PyExc_ValueError, "rng state does not look like a CPU mt19937 state");
        return -1;
    }
    *out = s;
    return 0;
}

/* randperm_check(state, n) -> list: the permutation this file would draw (advances `state` in place) */
static PyObject* py_randperm_check(PyObject* self, PyObject* args) {
    Py_buffer sb;
    long long n;
    if (!PyArg_ParseTuple(args, "w*L", &sb, &n)) return NULL;
    MtState* s;
    if (state_from_buffer(&sb, &s) < 0 || n < 0 || n > (1 << 24)) {
        if (!PyErr_Occurred()) PyErr_SetString(PyExc_ValueError, "bad n");
        PyBuffer_Release(&sb);
        return NULL;
    }
    int64_t* r = (int64_t*)PyMem_Malloc(sizeof(int64_t) * (size_t)(n ? n : 1));
    if (!r) { PyBuffer_Release(&sb); return PyErr_NoMemory(); }
    randperm(s, n, r);
    PyObject* out = PyList_New((Py_ssize_t)n);
    for (long long i = 0; out && i < n; ++i) {
        PyObject* v = PyLong_FromLongLong(r[i]);
        if (!v) { Py_CLEAR(out); break; }
        PyList_SET_ITEM(out, (Py_ssize_t)i, v);
    }
    PyMem_Free(r);
    PyBuffer_Release(&sb);
    return out;
}

/* Iteration order of `slots - set()` per class, kept between calls (draws without exclusions: two of the three draws of an ASER
 * step).  The order of the NEW set only depends on the contents and history of `slots`; the Python side keeps a version counter per
 * label (bumped whenever update_cache moves a slot in or out of the class) and a token that changes whenever the dict itself is
 * replaced, so an entry is reused only for an unchanged class of the same dict.  The set's size is compared as well. */
#define MEMO_LABELS 4096
typedef struct {
    long long token, version;
    Py_ssize_t n;
    uint64_t content;   /* checksum of the set's (element, table slot) pairs (set_checksum) when the order was recorded */
    int64_t* members;
} ClassMemo;
static ClassMemo g_memo[MEMO_LABELS];

/* The memo is keyed on (dict token, per-label version, set size); versions are bumped only by update_cache.  Any other in-place
 * mutation of a class set (a test, a future plugin, a remove + add of equal size) is caught by comparing a checksum of the live
 * set's (element hash, hash-table slot) pairs (one pass over the hash table: no set copy, no PyLong conversion): it changes when
 * the contents change AND when the same elements sit in other slots of the table, i.e. whenever the iteration order -- which is
 * what the memo stores -- can differ. */
static uint64_t set_checksum(PyObject* set) {
    Py_ssize_t pos = 0;
    PyObject* item;
    Py_hash_t h;
    uint64_t acc = 0;
    while (_PySet_NextEntry(set, &pos, &item, &h)) {
        uint64_t x = (uint64_t)h * 0x9E3779B97F4A7C15ull + (uint64_t)pos * 0xD6E8FEB86659FD93ull;   /* pos: one past the entry's slot */
        x ^= x >> 29;
        acc += x * 0xBF58476D1CE4E5B9ull;
    }
    return acc;
}

/* Iteration order of the NEW set `slots - excluded` without building it (round 6: the draw with exclusions was 220 - 320 us of host time on
 * the ASER step's critical path, most of it allocating and filling one Python set per class, and grew with the step count as the class sets'
 * tables did: profiles/r6_aser_drift_probe.txt).  CPython's set_difference, in the branch it takes when len(slots) / 4 <= len(excluded),
 * walks `slots` in table order and adds every element that is not in `excluded` to a fresh set; the fresh set's iteration order is the order
 * of its hash table.  That table is simulated here for non-negative ints below 2^61 (hash == value): first slot hash & mask, nine linear
 * probes, then i = 5 i + 1 + (perturb >>= 5); growth when 5 fill >= 3 mask to the power of two above 4 used, re-inserting the old table in
 * order (Objects/setobject.c: set_add_entry, set_table_resize, set_insert_clean; no dummies and no equal keys can occur in the fresh set).
 * Anything else -- other element types, other size ratios, subclasses -- returns -1 and the caller performs the real Python operation.
 * `setdiff_check` hands the simulated order to the Python side, which compares it with list(a - b) on churned sets before trusting it
 * (plugins/buffer_utils.py), as it does for randperm. */
#define EMU_LINEAR_PROBES 9
#define EMU_PERTURB_SHIFT 5
#define EMU_MAX_TABLE 16384
static int64_t g_emu_a[EMU_MAX_TABLE], g_emu_b[EMU_MAX_TABLE];
static inline void emu_place(int64_t* table, size_t mask, int64_t v) {
    size_t perturb = (size_t)v, i = (size_t)v & mask;
    for (;;) {
        if (table[i] < 0) { table[i] = v; return; }
        if (i + EMU_LINEAR_PROBES <= mask) {
            for (size_t j = 1; j <= EMU_LINEAR_PROBES; ++j)
                if (table[i + j] < 0) { table[i + j] = v; return; }
        }
        perturb >>= EMU_PERTURB_SHIFT;
        i = (i * 5 + 1 + perturb) & mask;
    }
}
/* -> number of members written (in the fresh set's iteration order), or -1: not simulated */
/* exmap (optional): exmap[v] != 0 <=> v is in `other`, for 0 <= v < exmap_n (every element of `other` lies below exmap_n): membership
 * without hashing a PyLong per element (5000 PySet_Contains calls were most of the simulated draw) */
static Py_ssize_t emu_difference(PyObject* so, PyObject* other, int64_t* members, Py_ssize_t cap, const unsigned char* exmap, Py_ssize_t exmap_n) {
    if (!PySet_CheckExact(so) || !PyAnySet_CheckExact(other)) return -1;
    if ((PySet_GET_SIZE(so) >> 2) > PySet_GET_SIZE(other)) return -1;   /* CPython copies `so` and discards instead */
    int64_t *cur = g_emu_a, *nxt = g_emu_b;
    size_t mask = 7, fill = 0;
    for (size_t k = 0; k <= mask; ++k) cur[k] = -1;
    Py_ssize_t pos = 0;
    PyObject* key;
    Py_hash_t h;
    while (_PySet_NextEntry(so, &pos, &key, &h)) {
        if (!PyLong_CheckExact(key) || h < 0 || h >= ((Py_hash_t)1 << 60)) return -1;   /* hash == value only for these */
        if (exmap) {
            if (h < exmap_n && exmap[h]) continue;
        } else {
            const int rv = PySet_Contains(other, key);
            if (rv < 0) { PyErr_Clear(); return -1; }
            if (rv) continue;
        }
        emu_place(cur, mask, (int64_t)h);
        ++fill;
        if (fill * 5 >= mask * 3) {
            const size_t minused = fill > 50000 ? fill * 2 : fill * 4;
            size_t newsize = 8;
            while (newsize <= minused) newsize <<= 1;
            if (newsize > EMU_MAX_TABLE) return -1;
            for (size_t k = 0; k < newsize; ++k) nxt[k] = -1;
            for (size_t k = 0; k <= mask; ++k)
                if (cur[k] >= 0) emu_place(nxt, newsize - 1, cur[k]);
            int64_t* t = cur; cur = nxt; nxt = t;
            mask = newsize - 1;
        }
    }
    if ((Py_ssize_t)fill > cap) return -1;
    Py_ssize_t n = 0;
    for (size_t k = 0; k <= mask; ++k)
        if (cur[k] >= 0) members[n++] = cur[k];
    return n;
}

/* setdiff_check(a: set, b: set) -> list | None: the simulated iteration order of `a - b` (None: this pair is not simulated) */
static PyObject* py_setdiff_check(PyObject* self, PyObject* args) {
    PyObject *a, *b;
    if (!PyArg_ParseTuple(args, "OO", &a, &b)) return NULL;
    int64_t* members = (int64_t*)PyMem_Malloc(sizeof(int64_t) * EMU_MAX_TABLE);
    if (!members) return PyErr_NoMemory();
    const Py_ssize_t n = emu_difference(a, b, members, EMU_MAX_TABLE, NULL, 0);
    PyObject* out;
    if (n < 0) {
        out = Py_None;
        Py_INCREF(out);
    } else {
        out = PyList_New(n);
        for (Py_ssize_t i = 0; out && i < n; ++i) PyList_SET_ITEM(out, i, PyLong_FromLongLong(members[i]));
    }
    PyMem_Free(members);
    return out;
}

/* cbrs_sample(class_index_cache: dict[label -> set[int]], excluded: set | None, n_smp_cls: int, state, out[, versions, token[, verify]])
 * -> number of picks.  out: writable buffer of int64; raises if it is too small.  versions: int64 buffer indexed by label.
 * verify (default 1): compare the live set's checksum before a memoised order is used.  The comparison walks the set's whole hash table --
 * 256 - 512 entries for ~50 members once update_cache has churned the sets for a few hundred steps -- and was most of the call by then
 * (profiles/r6_cbrs_drift_cpu.txt: 163 -> 279 us per draw over 1600 steps with it, 68 -> 85 us without); a caller whose every mutation goes
 * through update_cache (the ASER plugins) passes 0 and verifies now and then.
 * emulate (default 0): a draw with exclusions takes the order of `slots - excluded` from emu_difference where that applies. */
static PyObject* py_cbrs_sample(PyObject* self, PyObject* args) {
    PyObject *cache, *excluded;
    long long n_smp, token = 0;
    int verify = 1, emulate = 0;
    unsigned char* exmap = NULL;
    Py_ssize_t exmap_n = 0;
    Py_buffer sb, ob, vb;
    vb.buf = NULL; vb.obj = NULL; vb.len = 0;
    if (!PyArg_ParseTuple(args, "O!OLw*w*|y*Lii", &PyDict_Type, &cache, &excluded, &n_smp, &sb, &ob, &vb, &token, &verify, &emulate)) return NULL;
    MtState* s;
    PyObject* result = NULL;
    PyObject* empty = NULL;
    int64_t *members = NULL, *perm = NULL;
    Py_ssize_t cap = 0;
    if (state_from_buffer(&sb, &s) < 0) goto done;
    if (excluded == Py_None) {
        empty = PySet_New(NULL);
        if (!empty) goto done;
        excluded = empty;
    } else if (!PyAnySet_Check(excluded)) {
        PyErr_SetString(PyExc_TypeError, "excluded must be a set or None");
        goto done;
    }
    if (n_smp < 0) n_smp = 0;
    if (emulate && PyAnySet_CheckExact(excluded) && PySet_GET_SIZE(excluded) > 0) {   /* membership map of the excluded slots */
        Py_ssize_t pos = 0, top = -1;
        PyObject* item;
        Py_hash_t h;
        int plain = 1;
        while (_PySet_NextEntry(excluded, &pos, &item, &h)) {
            if (!PyLong_CheckExact(item) || h < 0 || h >= (1 << 24)) { plain = 0; break; }
            if (h > top) top = h;
        }
        if (plain && top >= 0) {
            exmap = (unsigned char*)PyMem_Calloc((size_t)top + 1, 1);
            if (exmap) {
                exmap_n = top + 1;
                pos = 0;
                while (_PySet_NextEntry(excluded, &pos, &item, &h)) exmap[h] = 1;
            }
        }
    }
    {
        const int64_t* versions = (const int64_t*)vb.buf;
        const Py_ssize_t n_versions = vb.buf ? vb.len / (Py_ssize_t)sizeof(int64_t) : 0;
        const int memo_ok = versions != NULL && PySet_GET_SIZE(excluded) == 0;
        int64_t* out = (int64_t*)ob.buf;
        const Py_ssize_t out_cap = ob.len / (Py_ssize_t)sizeof(int64_t);
        Py_ssize_t n_out = 0, pos = 0;
        PyObject *key, *slots;
        while (PyDict_Next(cache, &pos, &key, &slots)) {
            if (!PyAnySet_Check(slots)) {
                PyErr_SetString(PyExc_TypeError, "class_index_cache values must be sets");
                goto done;
            }
            if (PySet_GET_SIZE(slots) == 0) continue;
            ClassMemo* memo = NULL;
            long long memo_version = 0;
            if (memo_ok && PyLong_Check(key)) {
                const long long label = PyLong_AsLongLong(key);
                if (label == -1 && PyErr_Occurred()) PyErr_Clear();   /* a label beyond 64 bits: simply not memoised */
                if (label >= 0 && label < MEMO_LABELS && label < n_versions) {
                    memo = &g_memo[label];
                    if (memo->members && memo->token == token && memo->version == versions[label] && memo->n == PySet_GET_SIZE(slots) &&
                        (!verify || memo->content == set_checksum(slots))) {
                        const Py_ssize_t n = memo->n;
                        if (n > cap) {
                            cap = n * 2 + 64;
                            int64_t* m2 = (int64_t*)PyMem_Realloc(members, sizeof(int64_t) * (size_t)cap);
                            int64_t* p2 = m2 ? (int64_t*)PyMem_Realloc(perm, sizeof(int64_t) * (size_t)cap) : NULL;
                            if (m2) members = m2;
                            if (p2) perm = p2;
                            if (!m2 || !p2) { PyErr_NoMemory(); goto done; }
                        }
                        randperm(s, (int64_t)n, perm);
                        const Py_ssize_t take = n < (Py_ssize_t)n_smp ? n : (Py_ssize_t)n_smp;
                        if (n_out + take > out_cap) {
                            PyErr_SetString(PyExc_ValueError, "output buffer too small");
                            goto done;
                        }
                        for (Py_ssize_t j = 0; j < take; ++j) out[n_out++] = memo->members[perm[j]];
                        continue;
                    }
                    memo_version = versions[label];
                }
            }
            if (emulate && !memo) {   /* a draw with exclusions: the fresh set's order without the fresh set */
                const Py_ssize_t need = PySet_GET_SIZE(slots);
                if (need > cap) {
                    cap = need * 2 + 64;
                    int64_t* m2 = (int64_t*)PyMem_Realloc(members, sizeof(int64_t) * (size_t)cap);
                    int64_t* p2 = m2 ? (int64_t*)PyMem_Realloc(perm, sizeof(int64_t) * (size_t)cap) : NULL;
                    if (m2) members = m2;
                    if (p2) perm = p2;
                    if (!m2 || !p2) { PyErr_NoMemory(); goto done; }
                }
                const Py_ssize_t ne = emu_difference(slots, excluded, members, cap, exmap, exmap_n);
                if (ne >= 0) {
                    randperm(s, (int64_t)ne, perm);
                    const Py_ssize_t take = ne < (Py_ssize_t)n_smp ? ne : (Py_ssize_t)n_smp;
                    if (n_out + take > out_cap) {
                        PyErr_SetString(PyExc_ValueError, "output buffer too small");
                        goto done;
                    }
                    for (Py_ssize_t j = 0; j < take; ++j) out[n_out++] = members[perm[j]];
                    continue;
                }
            }
            PyObject* eligible = PyNumber_Subtract(slots, excluded);   /* a new set, exactly as `slots - excluded` */
            if (!eligible) goto done;
            const Py_ssize_t n = PySet_GET_SIZE(eligible);
            if (n > cap) {
                cap = n * 2 + 64;
                int64_t* m2 = (int64_t*)PyMem_Realloc(members, sizeof(int64_t) * (size_t)cap);
                int64_t* p2 = m2 ? (int64_t*)PyMem_Realloc(perm, sizeof(int64_t) * (size_t)cap) : NULL;
                if (m2) members = m2;
                if (p2) perm = p2;
                if (!m2 || !p2) { Py_DECREF(eligible); PyErr_NoMemory(); goto done; }
            }
            randperm(s, (int64_t)n, perm);                              /* drawn even when nothing is eligible (n == 0) */
            Py_ssize_t spos = 0, k = 0;
            PyObject* item;
            Py_hash_t h;
            while (_PySet_NextEntry(eligible, &spos, &item, &h)) {      /* CPython's own iteration order */
                const long long v = PyLong_AsLongLong(item);
                if (v == -1 && PyErr_Occurred()) { Py_DECREF(eligible); goto done; }
                members[k++] = (int64_t)v;
            }
            Py_DECREF(eligible);
            if (memo) {   /* keep the order for the next draw of this (unchanged) class */
                int64_t* mm = (int64_t*)PyMem_RawRealloc(memo->members, sizeof(int64_t) * (size_t)(n ? n : 1));
                if (mm) {   /* (version and token are stamped only with a complete entry) */
                    if (n) memcpy(mm, members, sizeof(int64_t) * (size_t)n);
                    memo->members = mm;
                    memo->n = n;
                    memo->content = set_checksum(slots);   /* (memo_ok: nothing is excluded, `eligible` has the elements of `slots`) */
                    memo->version = memo_version;
                    memo->token = token;
                } else {
                    PyMem_RawFree(memo->members);
                    memo->members = NULL;
                }
            }
            const Py_ssize_t take = n < (Py_ssize_t)n_smp ? n : (Py_ssize_t)n_smp;
            if (n_out + take > out_cap) {
                PyErr_SetString(PyExc_ValueError, "output buffer too small");
                goto done;
            }
            for (Py_ssize_t j = 0; j < take; ++j) out[n_out++] = members[perm[j]];
        }
        result = PyLong_FromSsize_t(n_out);
    }
done:
    PyMem_Free(exmap);
    PyMem_Free(members);
    PyMem_Free(perm);
    Py_XDECREF(empty);
    PyBuffer_Release(&sb);
    PyBuffer_Release(&ob);
    if (vb.obj) PyBuffer_Release(&vb);
    return result;
}

static PyMethodDef methods[] = {
    {"cbrs_sample", py_cbrs_sample, METH_VARARGS, "class-balanced draw (see file header)"},
    {"randperm_check", py_randperm_check, METH_VARARGS, "the permutation the restated generator draws"},
    {"setdiff_check", py_setdiff_check, METH_VARARGS, "the simulated iteration order of a - b (None: not simulated)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moddef = {PyModuleDef_HEAD_INIT, "_hostc", "host-side helpers (ASER class-balanced sampling)", -1, methods};

PyMODINIT_FUNC PyInit__hostc(void) { return PyModule_Create(&moddef); }
