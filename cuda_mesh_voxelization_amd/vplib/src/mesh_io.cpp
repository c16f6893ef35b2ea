// mesh_io.cpp -- OBJ reader/writer with the accepted subset and the output format of the reference
// (/root/reference/vplib/src/mesh/mesh_io.cpp:15-132): "v x y z [r g b]", "vn x y z",
// "f a//b c//d e//f" with 1-based indices; "# Vertices: n" / "# Faces: n" comments pre-reserve.
#include "mesh/mesh_io.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <filesystem>
#include <fstream>
#include <vector>

#include "debug_utils.h"

namespace {

// next whitespace-delimited token of [p, end); returns false at end of line
bool next_token(const char*& p, const char* end, const char*& tb, const char*& te)
{
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    if (p >= end) return false;
    tb = p;
    while (p < end && *p != ' ' && *p != '\t' && *p != '\r') ++p;
    te = p;
    return true;
}

float to_float(const char* b, const char* e)
{
    char buf[64];
    const size_t n = std::min<size_t>(e - b, sizeof(buf) - 1);
    std::memcpy(buf, b, n);
    buf[n] = 0;
    return std::strtof(buf, nullptr);           // what std::stof does underneath (mesh_io.cpp:55)
}

}  // namespace

// ---- binary cache of parsed meshes -------------------------------------------------------------------------
// Parsing a 10-million-triangle OBJ (about 1 GB of text) takes tens of seconds; the five arrays of the parsed Mesh are
// ~400 MB and load in a fraction of a second.  Opt-in: VPLIB_MESH_CACHE=1 in the environment (the reference has no such
// thing; its CLI re-parses on every run).  The cache file "<obj>.vpmesh" is keyed on the source's size and modification
// time and ignored when either differs or the file is malformed.
namespace {

struct CacheHeader {
    char magic[8];               // "VPMESH1\0"
    uint64_t srcSize;
    int64_t srcMtime;
    uint64_t counts[5];          // FacesCoords, FacesNormals, Coords, Normals, Colors (elements)
};

bool source_stamp(const std::string& filename, uint64_t& size, int64_t& mtime)
{
    std::error_code ec;
    const auto sz = std::filesystem::file_size(filename, ec);
    if (ec) return false;
    const auto tm = std::filesystem::last_write_time(filename, ec);
    if (ec) return false;
    size = static_cast<uint64_t>(sz);
    mtime = static_cast<int64_t>(tm.time_since_epoch().count());
    return true;
}

template <typename V>
bool read_vec(std::FILE* f, V& v, uint64_t count)
{
    v.resize(count);
    return count == 0 || std::fread(v.data(), sizeof(typename V::value_type), count, f) == count;
}

template <typename V>
bool write_vec(std::FILE* f, const V& v)
{
    return v.empty() || std::fwrite(v.data(), sizeof(typename V::value_type), v.size(), f) == v.size();
}

bool load_cache(const std::string& filename, Mesh& mesh)
{
    uint64_t size; int64_t mtime;
    if (!source_stamp(filename, size, mtime)) return false;
    std::FILE* f = std::fopen((filename + ".vpmesh").c_str(), "rb");
    if (!f) return false;
    CacheHeader h{};
    bool ok = std::fread(&h, sizeof(h), 1, f) == 1 && std::memcmp(h.magic, "VPMESH1", 8) == 0 && h.srcSize == size && h.srcMtime == mtime;
    if (ok) {
        // the counts of a corrupt or foreign file must not reach resize(): the payload they announce has to be exactly what
        // the file holds (overflow-safe: every count is bounded by the file size first), face indices come in threes
        std::error_code ec;
        const uint64_t fileSize = static_cast<uint64_t>(std::filesystem::file_size(filename + ".vpmesh", ec));
        const uint64_t elem[5] = {sizeof(mesh.FacesCoords[0]), sizeof(mesh.FacesNormals[0]), sizeof(mesh.Coords[0]), sizeof(mesh.Normals[0]), sizeof(mesh.Colors[0])};
        uint64_t payload = 0;
        ok = !ec && fileSize >= sizeof(h);
        for (int i = 0; ok && i < 5; ++i) {
            ok = h.counts[i] <= (fileSize - sizeof(h)) / elem[i];
            payload += h.counts[i] * elem[i];
        }
        ok = ok && payload == fileSize - sizeof(h) && h.counts[0] % 3 == 0;
    }
    if (ok) {
        mesh.Clear();
        try {
            ok = read_vec(f, mesh.FacesCoords, h.counts[0]) && read_vec(f, mesh.FacesNormals, h.counts[1]) && read_vec(f, mesh.Coords, h.counts[2]) &&
                 read_vec(f, mesh.Normals, h.counts[3]) && read_vec(f, mesh.Colors, h.counts[4]) && std::fgetc(f) == EOF;
        } catch (const std::exception&) {                           // bad_alloc / length_error: fall back to the parser
            ok = false;
        }
    }
    std::fclose(f);
    if (!ok) mesh.Clear();
    return ok;
}

void store_cache(const std::string& filename, const Mesh& mesh)
{
    CacheHeader h{};
    std::memcpy(h.magic, "VPMESH1", 8);
    if (!source_stamp(filename, h.srcSize, h.srcMtime)) return;
    h.counts[0] = mesh.FacesCoords.size(); h.counts[1] = mesh.FacesNormals.size(); h.counts[2] = mesh.Coords.size();
    h.counts[3] = mesh.Normals.size(); h.counts[4] = mesh.Colors.size();
    const std::string tmp = filename + ".vpmesh.tmp";
    std::FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return;                                                 // read-only directory: just no cache
    const bool ok = std::fwrite(&h, sizeof(h), 1, f) == 1 && write_vec(f, mesh.FacesCoords) && write_vec(f, mesh.FacesNormals) &&
                    write_vec(f, mesh.Coords) && write_vec(f, mesh.Normals) && write_vec(f, mesh.Colors);
    std::fclose(f);
    std::error_code ec;
    if (ok) std::filesystem::rename(tmp, filename + ".vpmesh", ec);
    if (!ok || ec) std::filesystem::remove(tmp, ec);
}

bool cache_enabled()
{
    const char* e = std::getenv("VPLIB_MESH_CACHE");
    return e && *e && std::strcmp(e, "0") != 0;
}

}  // namespace

static bool ParseObj(const std::string& filename, Mesh& mesh);

bool ImportMesh(const std::string filename, Mesh& mesh)
{
    const std::string ext = std::filesystem::path(filename).extension().string();
    if (ext != ".obj" && ext != ".OBJ") {
        LOG_ERROR("%s is a wrong file extension. It must be .obj or .OBJ", ext.c_str());
        return false;
    }
    if (cache_enabled() && load_cache(filename, mesh)) {
        mesh.Name = filename;
        return true;
    }
    if (!ParseObj(filename, mesh)) return false;
    if (cache_enabled()) store_cache(filename, mesh);
    return true;
}

static bool ParseObj(const std::string& filename, Mesh& mesh)
{
    std::ifstream file(filename, std::ios::binary | std::ios::ate);
    if (!file.is_open()) {
        LOG_ERROR("Error to open file %s", filename.c_str());
        return false;
    }
    const std::streamsize size = file.tellg();
    file.seekg(0);
    std::vector<char> text((size_t)size);
    if (size > 0 && !file.read(text.data(), size)) return false;

    mesh.Clear();
    const char* p = text.data();
    const char* const end = p + text.size();
    while (p < end) {
        const char* eol = static_cast<const char*>(std::memchr(p, '\n', end - p));
        if (!eol) eol = end;
        const char* q = p;
        const char *tb, *te;
        if (next_token(q, eol, tb, te)) {
            const size_t len = te - tb;
            if (len == 1 && *tb == '#') {
                int count = 0;
                const std::string line(p, eol);
                if (std::sscanf(line.c_str(), "# Vertices: %d", &count) == 1) mesh.VerticesReserve(count);
                else if (std::sscanf(line.c_str(), "# Faces: %d", &count) == 1) mesh.FacesReserve(count);
            } else if (len == 2 && tb[0] == 'v' && tb[1] == 'n') {
                float n[3] = {0, 0, 0};
                for (int i = 0; i < 3 && next_token(q, eol, tb, te); ++i) n[i] = to_float(tb, te);
                mesh.Normals.emplace_back(n[0], n[1], n[2]);
            } else if (len == 1 && *tb == 'v') {
                float c[6] = {0, 0, 0, 0, 0, 0};
                int got = 0;
                while (got < 6 && next_token(q, eol, tb, te)) c[got++] = to_float(tb, te);
                mesh.Coords.emplace_back(c[0], c[1], c[2]);
                if (got == 6) mesh.Colors.emplace_back(c[3], c[4], c[4], 1.0f);   // (r, g, g): reference quirk, mesh_io.cpp:59
            } else if (len == 1 && *tb == 'f') {
                for (int i = 0; i < 3; ++i) {
                    if (!next_token(q, eol, tb, te)) break;
                    char buf[64];
                    const size_t n = std::min<size_t>(te - tb, sizeof(buf) - 1);
                    std::memcpy(buf, tb, n);
                    buf[n] = 0;
                    unsigned pos = 0, nrm = 0;
                    std::sscanf(buf, " %u//%u", &pos, &nrm);
                    mesh.FacesCoords.push_back(pos - 1);
                    mesh.FacesNormals.push_back(nrm - 1);
                }
            }
        }
        p = eol + 1;
    }
    mesh.Name = filename;
    mesh.ShrinkToFit();
    return true;
}

bool ExportMesh(const std::string filename, const Mesh& mesh)
{
    std::FILE* f = std::fopen(filename.c_str(), "w");
    if (!f) {
        LOG_ERROR("Error to create or open %s file", filename.c_str());
        return false;
    }
    std::fprintf(f, "# OBJ file exporter (vplib, MI355X build)\n# Vertices: %zu\n# Faces: %zu\n", mesh.VerticesSize(), mesh.FacesSize());
    for (size_t i = 0; i < mesh.VerticesSize(); ++i) {
        const Color c = i < mesh.Colors.size() ? mesh.Colors[i] : Color();
        std::fprintf(f, "v %.6f %.6f %.6f %.6f %.6f %.6f\n", mesh.Coords[i].X, mesh.Coords[i].Y, mesh.Coords[i].Z,
                     c.R() / 255.0f, c.G() / 255.0f, c.B() / 255.0f);
    }
    std::fprintf(f, "\n");
    for (const Normal& n : mesh.Normals) std::fprintf(f, "vn %.6f %.6f %.6f\n", n.X, n.Y, n.Z);
    std::fprintf(f, "\n");
    const bool hasN = mesh.FacesNormals.size() == mesh.FacesCoords.size();
    for (size_t i = 0; i + 2 < mesh.FacesCoords.size(); i += 3) {
        const uint32_t* c = &mesh.FacesCoords[i];
        const uint32_t n0 = hasN ? mesh.FacesNormals[i] : c[0], n1 = hasN ? mesh.FacesNormals[i + 1] : c[1],
                       n2 = hasN ? mesh.FacesNormals[i + 2] : c[2];
        std::fprintf(f, "f %u//%u %u//%u %u//%u\n", c[0] + 1, n0 + 1, c[1] + 1, n1 + 1, c[2] + 1, n2 + 1);
    }
    std::fclose(f);
    LOG_INFO("Mesh %s sucessfully exported", filename.c_str());
    return true;
}
