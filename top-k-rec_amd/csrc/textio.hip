// Host-side readers/writers of the reference's text formats (no GPU work in this file).
//
// once K4 scores a catalogue in milliseconds the pure-Python parsers of the reference are
// the whole wall time of evaluate.py (get_history 12.6 s + get_mat 2.6 s on ML-10M) and of load_training_data.
// These entry points parse a ratings file or a '%f ' matrix in one pass over the mapped file and hand back flat
// arrays; the host mirror (utils.py / evaluate.py / single/bpr.py) builds the reference's dicts from them.
//
//   ratings line   "uid,iid:like,iid:like,..."   utils.py:58-70, evaluate.py:30-45 and :84-93
//   matrix line    "%f %f ... %f \n"             utils.py:28-44 (read), :47-55 (write)
//
// Restated semantics: a line is what Python's `for line in open(path)` yields with '\n' separators, stripped of
// ASCII whitespace at both ends; fields are split on ',' and ':' without further stripping; `like` is the integer
// between the first and the second ':' of a field (evaluate.py:38 int(...)); a field without ':' is an error in the
// reference (IndexError) and TKR_E_PARSE here.  Matrix elements are parsed with strtod and narrowed to fp32 --
// the same two roundings as np.float32(str) -- and written with "%f " (printf and Python's '%f' agree on every
// finite double).
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../include/tkr.h"

namespace {

struct IdMap {
    std::string blob;                                           // owns the token bytes
    std::unordered_map<std::string_view, int32_t> table;
};

struct Ratings {
    std::vector<int32_t> line_user;                             // index of the line's uid, -1 = not in the map
    std::vector<int64_t> line_ptr;                              // entries of line l: [line_ptr[l], line_ptr[l+1])
    std::vector<int32_t> item;                                  // index of the field's iid, -1 = not in the map
    std::vector<int32_t> like;
};

struct Matrix {
    std::vector<float> data;
    int64_t rows = 0, cols = 0;
};

struct Mapped {
    const char* p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        if (n == 0) { p = ""; return true; }
        void* m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = static_cast<const char*>(m);
        return true;
    }
    ~Mapped() {
        if (p && n) munmap(const_cast<char*>(p), n);
        if (fd >= 0) close(fd);
    }
};

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r' || c == '\v' || c == '\f'; }

// [b, e) -> stripped
inline void strip(const char*& b, const char*& e) {
    while (b < e && is_space(*b)) ++b;
    while (e > b && is_space(e[-1])) --e;
}

// Python int(str): optional whitespace, sign, decimal digits (underscores and non-ASCII digits not handled)
inline bool parse_int(const char* b, const char* e, int32_t& out) {
    strip(b, e);
    if (b == e) return false;
    bool neg = false;
    if (*b == '+' || *b == '-') { neg = (*b == '-'); ++b; }
    if (b == e) return false;
    int64_t v = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        v = v * 10 + (*b - '0');
        if (v > 2147483647LL) v = 2147483647LL;
    }
    out = (int32_t)(neg ? -v : v);
    return true;
}

}  // namespace

extern "C" int tkr_idmap_create(const char* blob, int64_t blob_len, const int32_t* index, int64_t n, void** out) {
    try {
        if (!out || n < 0 || blob_len < 0 || (n > 0 && (!blob || !index))) return TKR_E_INVAL;
        std::unique_ptr<IdMap> m(new IdMap());
        m->blob.assign(blob ? blob : "", (size_t)blob_len);
        m->table.reserve((size_t)n * 2);
        const char* p = m->blob.data();
        const char* end = p + m->blob.size();
        int64_t k = 0;
        while (k < n) {                                             // n tokens separated by '\n' (the last one unterminated)
            const char* q = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
            if (!q) q = end;
            m->table[std::string_view(p, (size_t)(q - p))] = index[k++];
            if (q == end) break;
            p = q + 1;
        }
        if (k != n) return TKR_E_INVAL;
        *out = m.release();
        return TKR_OK;
    } catch (...) {
        return TKR_E_NOMEM;                                       // nothing throws across the C ABI
    }
}

extern "C" int tkr_idmap_destroy(void* map) {
    delete static_cast<IdMap*>(map);
    return TKR_OK;
}

extern "C" int tkr_ratings_parse(const char* path, const void* users, const void* items, void** out) {
    try {
        if (!path || !users || !items || !out) return TKR_E_INVAL;
        const IdMap* um = static_cast<const IdMap*>(users);
        const IdMap* im = static_cast<const IdMap*>(items);
        Mapped f;
        if (!f.open(path)) return TKR_E_IO;
        std::unique_ptr<Ratings> r(new Ratings());
        r->line_ptr.push_back(0);
        const char* p = f.p;
        const char* end = f.p + f.n;
        while (p < end) {
            const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
            const char* le = nl ? nl : end;
            const char* b = p;
            const char* e = le;
            strip(b, e);
            // uid = up to the first ','
            const char* c = static_cast<const char*>(memchr(b, ',', (size_t)(e - b)));
            const char* ue = c ? c : e;
            auto it = um->table.find(std::string_view(b, (size_t)(ue - b)));
            r->line_user.push_back(it == um->table.end() ? -1 : it->second);
            const char* q = c ? c + 1 : e;
            while (c) {                                             // one field per ','
                const char* nc = static_cast<const char*>(memchr(q, ',', (size_t)(e - q)));
                const char* fe = nc ? nc : e;
                const char* colon = static_cast<const char*>(memchr(q, ':', (size_t)(fe - q)));
                if (!colon) return TKR_E_PARSE;                     // terms[k].split(':')[1] -> IndexError in the reference
                const char* l0 = colon + 1;
                const char* colon2 = static_cast<const char*>(memchr(l0, ':', (size_t)(fe - l0)));
                const char* l1 = colon2 ? colon2 : fe;
                int32_t like = 0;
                if (!parse_int(l0, l1, like)) return TKR_E_PARSE;
                auto jt = im->table.find(std::string_view(q, (size_t)(colon - q)));
                r->item.push_back(jt == im->table.end() ? -1 : jt->second);
                r->like.push_back(like);
                c = nc;
                q = nc ? nc + 1 : e;
            }
            r->line_ptr.push_back((int64_t)r->item.size());
            p = nl ? nl + 1 : end;
        }
        *out = r.release();
        return TKR_OK;
    } catch (...) {
        return TKR_E_NOMEM;                                       // nothing throws across the C ABI
    }
}

extern "C" int tkr_ratings_sizes(const void* ratings, int64_t* n_lines, int64_t* n_entries) {
    if (!ratings || !n_lines || !n_entries) return TKR_E_INVAL;
    const Ratings* r = static_cast<const Ratings*>(ratings);
    *n_lines = (int64_t)r->line_user.size();
    *n_entries = (int64_t)r->item.size();
    return TKR_OK;
}

extern "C" int tkr_ratings_copy(const void* ratings, int32_t* line_user, int64_t* line_ptr, int32_t* item, int32_t* like) {
    if (!ratings || !line_user || !line_ptr || !item || !like) return TKR_E_INVAL;
    const Ratings* r = static_cast<const Ratings*>(ratings);
    memcpy(line_user, r->line_user.data(), r->line_user.size() * sizeof(int32_t));
    memcpy(line_ptr, r->line_ptr.data(), r->line_ptr.size() * sizeof(int64_t));
    memcpy(item, r->item.data(), r->item.size() * sizeof(int32_t));
    memcpy(like, r->like.data(), r->like.size() * sizeof(int32_t));
    return TKR_OK;
}

extern "C" int tkr_ratings_destroy(void* ratings) {
    delete static_cast<Ratings*>(ratings);
    return TKR_OK;
}

extern "C" int tkr_matrix_read(const char* path, void** out) {
    try {
        if (!path || !out) return TKR_E_INVAL;
        Mapped f;
        if (!f.open(path)) return TKR_E_IO;
        std::unique_ptr<Matrix> m(new Matrix());
        const char* p = f.p;
        const char* end = f.p + f.n;
        std::string tok;
        while (p < end) {
            const char* nl = static_cast<const char*>(memchr(p, '\n', (size_t)(end - p)));
            const char* b = p;
            const char* e = nl ? nl : end;
            strip(b, e);
            int64_t cols = 0;
            while (b < e) {                                         // terms = line.strip().split(' ')
                const char* sp = static_cast<const char*>(memchr(b, ' ', (size_t)(e - b)));
                const char* te = sp ? sp : e;
                tok.assign(b, (size_t)(te - b));
                char* stop = nullptr;
                errno = 0;
                const double v = strtod(tok.c_str(), &stop);        // float(str) then narrowing, like np.float32(str)
                if (tok.empty() || stop != tok.c_str() + tok.size()) return TKR_E_PARSE;
                m->data.push_back((float)v);
                ++cols;
                b = sp ? sp + 1 : e;
            }
            if (m->rows == 0) m->cols = cols;
            if (cols != m->cols) return TKR_E_PARSE; // ragged rows: numpy raises on the assignment
            ++m->rows;
            p = nl ? nl + 1 : end;
        }
        *out = m.release();
        return TKR_OK;
    } catch (...) {
        return TKR_E_NOMEM;                                       // nothing throws across the C ABI
    }
}

extern "C" int tkr_matrix_sizes(const void* matrix, int64_t* rows, int64_t* cols) {
    if (!matrix || !rows || !cols) return TKR_E_INVAL;
    const Matrix* m = static_cast<const Matrix*>(matrix);
    *rows = m->rows;
    *cols = m->cols;
    return TKR_OK;
}

extern "C" int tkr_matrix_copy(const void* matrix, float* dst) {
    if (!matrix || !dst) return TKR_E_INVAL;
    const Matrix* m = static_cast<const Matrix*>(matrix);
    memcpy(dst, m->data.data(), m->data.size() * sizeof(float));
    return TKR_OK;
}

extern "C" int tkr_matrix_destroy(void* matrix) {
    delete static_cast<Matrix*>(matrix);
    return TKR_OK;
}

extern "C" int tkr_matrix_write(const char* path, const float* data, int64_t rows, int64_t cols) {
    try {
        if (!path || rows < 0 || cols < 0 || (rows * cols > 0 && !data)) return TKR_E_INVAL;
        FILE* fh = fopen(path, "w");
        if (!fh) return TKR_E_IO;
        std::vector<char> buf((size_t)cols * 52 + 2);               // "%f" of a float: at most 1 + 39 + 1 + 6 characters
        for (int64_t r = 0; r < rows; ++r) {
            char* w = buf.data();
            for (int64_t c = 0; c < cols; ++c) {
                const double v = (double)data[r * cols + c];
                if (isnan(v)) { memcpy(w, "nan ", 4); w += 4; }    // Python prints 'nan' for either sign
                else w += sprintf(w, "%f ", v);
            }
            *w++ = '\n';
            if (fwrite(buf.data(), 1, (size_t)(w - buf.data()), fh) != (size_t)(w - buf.data())) { fclose(fh); return TKR_E_IO; }
        }
        return fclose(fh) == 0 ? TKR_OK : TKR_E_IO;
    } catch (...) {
        return TKR_E_NOMEM;                                       // nothing throws across the C ABI
    }
}
