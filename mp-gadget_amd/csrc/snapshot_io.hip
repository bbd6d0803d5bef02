// snapshot_io.hip -- reader / writer of the reference's snapshot and IC wire format: "bigfile" blocks as
// petaio.c (libgadget/petaio.c:986-1120) lays them out, one directory per block (`0/Position`, `1/Velocity`, ..., `Header`).
// Host code only (file IO); it lets the engine consume MP-GenIC initial conditions / snapshots and write `PART` files.
// The format (depends/bigfile/src/bigfile.c):
//   <file>/<block>/header    text:  "DTYPE: <f8\nNMEMB: 3\nNFILE: 2\n" then per data file "%06X: size : checksum : sysv-sum"
//                            (bigfile.c:588-606; sizes in elements of NMEMB values; the checksum is the byte sum, :1420-1428)
//   <file>/<block>/%06X      raw little-endian data, element = NMEMB values of DTYPE (bigfile.c:893-962)
//   <file>/<block>/attr-v2   one attribute per line: "name dtype nmemb HEXBYTES #HUMANE [ text ]" (bigfile.c:1486-1627)
// Interoperability with the reference library is tested both ways (tests/test_snapshot_io.py builds bigfile.c in place).
#include "snapshot_io.h"
#include <cerrno>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/stat.h>
#include <vector>

namespace mpg {

namespace {

std::string normalize_dtype(const char *s) // _dtype_normalize, bigfile.c:989-1017 (little-endian machine)
{
    MPG_CHECK(s && *s, "bigfile: empty dtype");
    std::string d(s);
    if(d[0] != '<' && d[0] != '>' && d[0] != '|' && d[0] != '=')
        d = "=" + d;
    if(d[0] == '=' || d[0] == '|')
        d[0] = '<';
    MPG_CHECK(d.size() >= 3 && d.size() < 8, "bigfile: bad dtype `" + std::string(s) + "'");
    MPG_CHECK(d[0] == '<', "bigfile: big-endian data is not supported (`" + std::string(s) + "')");
    const char k = d[1];
    MPG_CHECK(k == 'f' || k == 'i' || k == 'u' || k == 'S' || k == 'b' || k == 'c', "bigfile: bad dtype kind in `" + std::string(s) + "'");
    const int w = atoi(d.c_str() + 2);
    MPG_CHECK(w > 0 && w <= 16, "bigfile: bad dtype width in `" + std::string(s) + "'");
    return d;
}
int itemsize(const std::string &nd) { return atoi(nd.c_str() + 2); }

template <typename T>
T load_as(const std::string &d, const unsigned char *p)
{
    const int w = itemsize(d);
    switch(d[1]) {
    case 'f':
        if(w == 8) {
            double v;
            memcpy(&v, p, 8);
            return (T)v;
        }
        if(w == 4) {
            float v;
            memcpy(&v, p, 4);
            return (T)v;
        }
        break;
    case 'i':
        if(w == 8) {
            int64_t v;
            memcpy(&v, p, 8);
            return (T)v;
        }
        if(w == 4) {
            int32_t v;
            memcpy(&v, p, 4);
            return (T)v;
        }
        if(w == 2) {
            int16_t v;
            memcpy(&v, p, 2);
            return (T)v;
        }
        if(w == 1)
            return (T)(int8_t)p[0];
        break;
    case 'u':
    case 'b':
    case 'S':
        if(w == 8) {
            uint64_t v;
            memcpy(&v, p, 8);
            return (T)v;
        }
        if(w == 4) {
            uint32_t v;
            memcpy(&v, p, 4);
            return (T)v;
        }
        if(w == 2) {
            uint16_t v;
            memcpy(&v, p, 2);
            return (T)v;
        }
        if(w == 1)
            return (T)p[0];
        break;
    }
    fail(__FILE__, __LINE__, "bigfile: unsupported dtype `" + d + "'");
}

template <typename T>
void store_from(const std::string &d, unsigned char *p, T v)
{
    const int w = itemsize(d);
    switch(d[1]) {
    case 'f':
        if(w == 8) {
            const double x = (double)v;
            memcpy(p, &x, 8);
            return;
        }
        if(w == 4) {
            const float x = (float)v;
            memcpy(p, &x, 4);
            return;
        }
        break;
    case 'i':
        if(w == 8) {
            const int64_t x = (int64_t)v;
            memcpy(p, &x, 8);
            return;
        }
        if(w == 4) {
            const int32_t x = (int32_t)v;
            memcpy(p, &x, 4);
            return;
        }
        if(w == 2) {
            const int16_t x = (int16_t)v;
            memcpy(p, &x, 2);
            return;
        }
        if(w == 1) {
            p[0] = (unsigned char)(int8_t)v;
            return;
        }
        break;
    case 'u':
    case 'b':
    case 'S':
        if(w == 8) {
            const uint64_t x = (uint64_t)v;
            memcpy(p, &x, 8);
            return;
        }
        if(w == 4) {
            const uint32_t x = (uint32_t)v;
            memcpy(p, &x, 4);
            return;
        }
        if(w == 2) {
            const uint16_t x = (uint16_t)v;
            memcpy(p, &x, 2);
            return;
        }
        if(w == 1) {
            p[0] = (unsigned char)v;
            return;
        }
        break;
    }
    fail(__FILE__, __LINE__, "bigfile: unsupported dtype `" + d + "'");
}

// dtype_convert_simple, bigfile.c:1100-1180: value conversion between the scalar types (floats through double, integers through 64 bits)
void convert(unsigned char *dst, const std::string &dd, const unsigned char *src, const std::string &sd, size_t n)
{
    const int dw = itemsize(dd), sw = itemsize(sd);
    if(dd == sd) {
        memcpy(dst, src, n * (size_t)dw);
        return;
    }
    const bool sf = sd[1] == 'f', df = dd[1] == 'f', ss = sd[1] == 'i';
    for(size_t k = 0; k < n; k++) {
        const unsigned char *s = src + k * (size_t)sw;
        unsigned char *d = dst + k * (size_t)dw;
        if(sf || df) {
            if(sf)
                store_from<double>(dd, d, load_as<double>(sd, s));
            else if(ss)
                store_from<double>(dd, d, (double)load_as<int64_t>(sd, s));
            else
                store_from<double>(dd, d, (double)load_as<uint64_t>(sd, s));
        }
        else if(ss)
            store_from<int64_t>(dd, d, load_as<int64_t>(sd, s));
        else
            store_from<uint64_t>(dd, d, load_as<uint64_t>(sd, s));
    }
}

std::string block_dir(const char *file, const char *block)
{
    MPG_CHECK(file && block && *file && *block, "bigfile: empty file or block name");
    MPG_CHECK(!strpbrk(block, " \t\n"), "bigfile: column name cannot contain blanks"); // bigfile.c:479-485
    return std::string(file) + "/" + block;
}

void mkdir_p(const std::string &path) // _big_file_mksubdir_r
{
    for(size_t i = 1; i <= path.size(); i++)
        if(i == path.size() || path[i] == '/') {
            const std::string sub = path.substr(0, i);
            if(mkdir(sub.c_str(), 0777) != 0 && errno != EEXIST)
                fail(__FILE__, __LINE__, "bigfile: cannot create directory `" + sub + "': " + strerror(errno));
        }
}

struct Header {
    std::string dtype;
    int nmemb = 0, nfile = 0;
    std::vector<size_t> fsize, foffset;
    std::vector<unsigned> cksum;
};

Header read_header(const std::string &dir)
{
    Header h;
    FILE *f = fopen((dir + "/header").c_str(), "r");
    MPG_CHECK(f != nullptr, "bigfile: cannot open `" + dir + "/header': " + strerror(errno));
    char dt[64];
    const bool ok = fscanf(f, " DTYPE: %63s", dt) == 1 && fscanf(f, " NMEMB: %d", &h.nmemb) == 1 && fscanf(f, " NFILE: %d", &h.nfile) == 1;
    if(!ok || h.nfile < 0 || h.nmemb < 0) {
        fclose(f);
        fail(__FILE__, __LINE__, "bigfile: failed to read the header of block `" + dir + "'");
    }
    h.dtype = normalize_dtype(dt);
    h.fsize.assign((size_t)h.nfile + 1, 0);
    h.cksum.assign((size_t)h.nfile + 1, 0);
    for(int i = 0; i < h.nfile; i++) {
        unsigned fid = 0, ck = 0, sysv = 0;
        size_t size = 0;
        if(fscanf(f, " %X: %zu : %u : %u", &fid, &size, &ck, &sysv) != 4 || (int)fid >= h.nfile) {
            fclose(f);
            fail(__FILE__, __LINE__, "bigfile: failed to read the physical file layout of `" + dir + "'");
        }
        h.fsize[fid] = size;
        h.cksum[fid] = ck;
    }
    fclose(f);
    h.foffset.assign((size_t)h.nfile + 1, 0);
    for(int i = 0; i < h.nfile; i++)
        h.foffset[i + 1] = h.foffset[i] + h.fsize[i];
    return h;
}

void write_header(const std::string &dir, const Header &h) // big_block_flush, bigfile.c:586-618
{
    FILE *f = fopen((dir + "/header").c_str(), "w+");
    MPG_CHECK(f != nullptr, "bigfile: cannot write `" + dir + "/header': " + strerror(errno));
    fprintf(f, "DTYPE: %s\nNMEMB: %d\nNFILE: %d\n", h.dtype.c_str(), h.nmemb, h.nfile);
    for(int i = 0; i < h.nfile; i++) {
        const unsigned s = h.cksum[i];
        const unsigned r = (s & 0xffff) + ((s & 0xffffffff) >> 16);
        const unsigned checksum = (r & 0xffff) + (r >> 16);
        fprintf(f, "%06X: %zu : %u : %u\n", (unsigned)i, h.fsize[i], h.cksum[i], checksum);
    }
    fclose(f);
}

std::string data_file(const std::string &dir, int fid)
{
    char b[16];
    snprintf(b, sizeof(b), "/%06X", (unsigned)fid);
    return dir + b;
}

struct Attr {
    std::string name, dtype;
    int nmemb;
    std::vector<unsigned char> data;
};

std::vector<Attr> read_attrs(const std::string &dir)
{
    std::vector<Attr> out;
    FILE *f = fopen((dir + "/attr-v2").c_str(), "r");
    if(!f)
        return out;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::string buf((size_t)(size > 0 ? size : 0), '\0');
    if(size > 0 && fread(&buf[0], 1, (size_t)size, f) != (size_t)size) {
        fclose(f);
        fail(__FILE__, __LINE__, "bigfile: failed to read `" + dir + "/attr-v2'");
    }
    fclose(f);
    size_t i = 0;
    auto blank = [](char c) { return c == ' ' || c == '\t' || c == '\n' || c == '\r'; };
    auto token = [&]() {
        while(i < buf.size() && blank(buf[i]))
            i++;
        const size_t b = i;
        while(i < buf.size() && !blank(buf[i]))
            i++;
        return buf.substr(b, i - b);
    };
    while(true) {
        const std::string name = token();
        if(name.empty())
            break;
        Attr a;
        a.name = name;
        a.dtype = normalize_dtype(token().c_str());
        a.nmemb = atoi(token().c_str());
        // "%s %s %d %s #HUMANE [ %s ]" (bigfile.c:1615): a zero-length attribute has an empty hex field, so the next token on its
        // line is the "#HUMANE" comment - not data
        std::string raw;
        if(a.nmemb > 0) {
            raw = token();
            if(!raw.empty() && raw[0] == '#')
                raw.clear();
        }
        while(i < buf.size() && buf[i] != '\n')
            i++;
        const size_t nbytes = (size_t)a.nmemb * (size_t)itemsize(a.dtype);
        MPG_CHECK(raw.size() == 2 * nbytes, "bigfile: NMEMB and data mismatch in attribute `" + name + "' of `" + dir + "'");
        a.data.resize(nbytes);
        for(size_t k = 0; k < nbytes; k++) {
            const char hx[3] = {raw[2 * k], raw[2 * k + 1], 0};
            a.data[k] = (unsigned char)strtol(hx, nullptr, 16);
        }
        out.push_back(a);
    }
    return out;
}

void write_attrs(const std::string &dir, const std::vector<Attr> &attrs) // attrset_write_attr_set_v2, bigfile.c:1560-1627
{
    FILE *f = fopen((dir + "/attr-v2").c_str(), "w");
    MPG_CHECK(f != nullptr, "bigfile: cannot write `" + dir + "/attr-v2': " + strerror(errno));
    static const char conv[] = "0123456789ABCDEF";
    for(const Attr &a : attrs) {
        std::string raw;
        for(unsigned char c : a.data) {
            raw.push_back(conv[c / 16]);
            raw.push_back(conv[c % 16]);
        }
        std::string text;
        if(a.data.size() > 128)
            text = "... (Too Long) ";
        else {
            const int w = itemsize(a.dtype);
            for(int j = 0; j < a.nmemb; j++) {
                const unsigned char *p = a.data.data() + (size_t)j * w;
                if(a.dtype[1] == 'S' && w == 1) {
                    if(p[0] == '\n') {
                        text += "...";
                        break;
                    }
                    if(p[0] == 0)
                        break;
                    text.push_back((char)p[0]);
                    continue;
                }
                char b[64];
                if(a.dtype[1] == 'f')
                    snprintf(b, sizeof(b), "%g", load_as<double>(a.dtype, p));
                else if(a.dtype[1] == 'i')
                    snprintf(b, sizeof(b), "%" PRId64, load_as<int64_t>(a.dtype, p));
                else
                    snprintf(b, sizeof(b), "%" PRIu64, load_as<uint64_t>(a.dtype, p));
                text += b;
                if(j != a.nmemb - 1)
                    text += " ";
            }
        }
        fprintf(f, "%s %s %d %s #HUMANE [ %s ]\n", a.name.c_str(), a.dtype.c_str(), a.nmemb, raw.c_str(), text.c_str());
    }
    fclose(f);
}

} // namespace

void bigfile_block_info(const char *file, const char *block, BigBlockInfo *info)
{
    const Header h = read_header(block_dir(file, block));
    memset(info->dtype, 0, sizeof(info->dtype));
    strncpy(info->dtype, h.dtype.c_str(), 7);
    info->nmemb = h.nmemb;
    info->nfile = h.nfile;
    info->size = (int64_t)h.foffset[h.nfile];
}

void bigfile_read_block(const char *file, const char *block, int64_t start, int64_t count, const char *want_dtype, void *out)
{
    const std::string dir = block_dir(file, block);
    const Header h = read_header(dir);
    const std::string want = normalize_dtype(want_dtype);
    const int64_t total = (int64_t)h.foffset[h.nfile];
    MPG_CHECK(start >= 0 && count >= 0 && start + count <= total, "bigfile: read beyond the end of block `" + dir + "'");
    const size_t fel = (size_t)itemsize(h.dtype) * (size_t)h.nmemb, wel = (size_t)itemsize(want) * (size_t)h.nmemb;
    std::vector<unsigned char> buf;
    unsigned char *dst = (unsigned char *)out;
    int64_t pos = start, left = count;
    for(int fid = 0; fid < h.nfile && left > 0; fid++) {
        const int64_t lo = (int64_t)h.foffset[fid], hi = (int64_t)h.foffset[fid + 1];
        if(pos >= hi)
            continue;
        const int64_t n = (hi - pos < left) ? hi - pos : left;
        FILE *f = fopen(data_file(dir, fid).c_str(), "r");
        MPG_CHECK(f != nullptr, "bigfile: cannot open `" + data_file(dir, fid) + "': " + strerror(errno));
        buf.resize((size_t)n * fel);
        const bool ok = fseek(f, (long)((pos - lo) * (int64_t)fel), SEEK_SET) == 0 && fread(buf.data(), fel, (size_t)n, f) == (size_t)n;
        fclose(f);
        MPG_CHECK(ok, "bigfile: short read in `" + data_file(dir, fid) + "'");
        convert(dst, want, buf.data(), h.dtype, (size_t)n * (size_t)h.nmemb);
        dst += (size_t)n * wel;
        pos += n;
        left -= n;
    }
}

void bigfile_write_block(const char *file, const char *block, const char *dtype, int nmemb, int nfile, int64_t size, const char *src_dtype,
                         const void *data)
{
    MPG_CHECK(nmemb >= 0 && nfile >= 0 && size >= 0 && (nfile > 0 || size == 0), "bigfile: bad block shape");
    const std::string dir = block_dir(file, block);
    mkdir_p(dir);
    Header h;
    h.dtype = normalize_dtype(dtype);
    const std::string sd = normalize_dtype(src_dtype ? src_dtype : dtype);
    h.nmemb = nmemb;
    h.nfile = nfile;
    h.fsize.assign((size_t)nfile + 1, 0);
    h.cksum.assign((size_t)nfile + 1, 0);
    for(int i = 0; i < nfile; i++) // the even split of big_file_create_block's callers (petaio.c:893-905)
        h.fsize[i] = (size_t)(size * (i + 1) / nfile - size * i / nfile);
    const size_t fel = (size_t)itemsize(h.dtype) * (size_t)nmemb, sel = (size_t)itemsize(sd) * (size_t)nmemb;
    const unsigned char *src = (const unsigned char *)data;
    std::vector<unsigned char> buf;
    for(int fid = 0; fid < nfile; fid++) {
        const size_t n = h.fsize[fid];
        buf.resize(n * fel);
        if(n > 0)
            convert(buf.data(), h.dtype, src, sd, n * (size_t)nmemb);
        unsigned sum = 0;
        for(unsigned char c : buf)
            sum += c; // sysvsum, bigfile.c:1420-1428
        h.cksum[fid] = sum;
        FILE *f = fopen(data_file(dir, fid).c_str(), "w");
        MPG_CHECK(f != nullptr, "bigfile: cannot write `" + data_file(dir, fid) + "': " + strerror(errno));
        const bool ok = n == 0 || fwrite(buf.data(), fel, n, f) == n;
        fclose(f);
        MPG_CHECK(ok, "bigfile: short write in `" + data_file(dir, fid) + "'");
        src += n * sel;
    }
    write_header(dir, h);
}

int bigfile_get_attr(const char *file, const char *block, const char *name, const char *want_dtype, void *out, int nmemb)
{
    const std::string dir = block_dir(file, block);
    const std::string want = normalize_dtype(want_dtype);
    for(const Attr &a : read_attrs(dir))
        if(a.name == name) {
            MPG_CHECK(a.nmemb == nmemb, "bigfile: attribute `" + std::string(name) + "' has a different number of members"); // bigfile.c:1750
            convert((unsigned char *)out, want, a.data.data(), a.dtype, (size_t)nmemb);
            return 0;
        }
    return 1; // no such attribute
}

void bigfile_set_attr(const char *file, const char *block, const char *name, const char *dtype, const void *data, int nmemb)
{
    MPG_CHECK(name && *name && !strpbrk(name, " \t\n"), "bigfile: attribute name cannot contain blanks");
    const std::string dir = block_dir(file, block);
    mkdir_p(dir);
    {   // a block that only carries attributes (`Header`) still has a header file (dtype i8, no data files; bigfile.c:492-497)
        struct stat st;
        if(stat((dir + "/header").c_str(), &st) != 0) {
            Header h;
            h.dtype = "<i8";
            h.fsize.assign(1, 0);
            h.cksum.assign(1, 0);
            write_header(dir, h);
        }
    }
    std::vector<Attr> attrs = read_attrs(dir);
    Attr a;
    a.name = name;
    a.dtype = normalize_dtype(dtype);
    a.nmemb = nmemb;
    a.data.assign((const unsigned char *)data, (const unsigned char *)data + (size_t)nmemb * (size_t)itemsize(a.dtype));
    bool found = false;
    for(Attr &b : attrs)
        if(b.name == a.name) {
            b = a;
            found = true;
        }
    if(!found) { // kept sorted by name like the reference's attribute set (bigfile.c:1660-1700)
        size_t k = 0;
        while(k < attrs.size() && attrs[k].name < a.name)
            k++;
        attrs.insert(attrs.begin() + (long)k, a);
    }
    write_attrs(dir, attrs);
}

} // namespace mpg
