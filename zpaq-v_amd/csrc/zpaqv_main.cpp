// zpaqv -- command-line front end with the reference CLI's commands, flags and messages
// (cmd/main.v:1-535): a/add, x/extract, l/list; -mN -force -test -to -only -not -sN.
// All files of one invocation are coded as ONE GPU batch (zpaq::archive_add / archive_extract);
// the archive bytes are what the reference's per-file loop writes (one block, one segment per
// file, basename as segment name, "<n> bytes" as comment; cmd/main.v:283-311).
// Flags the reference parses but never acts on (-all -index -key -noattributes -repack -threads -until)
// are accepted and ignored the same way.  -fragment N, which the reference also parses and ignores, is
// given its advertised meaning here as an opt-in: files are cut into 2^N KiB blocks.
#include <dirent.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>

#include <string>
#include <vector>

#include "../../include/zpaq_frontend.hpp"

namespace {

struct Config {
    std::string command, archive;
    std::vector<std::string> files, not_files, only_files, to_files;
    bool force = false, test_mode = false;
    int method = 1, summary = 0;
    int threads = 1;              // -tN / -threads N: GPUs to spread the blocks over (0 = all visible); the reference ignores it
    int fragment = -1;            // -fragment N given: cut files into 2^N KiB blocks (the reference parses and ignores it)
};

bool is_numeric(const std::string &s)
{
    if (s.empty()) return false;
    for (char c : s) if (c < '0' || c > '9') return false;
    return true;
}

// cmd/main.v:503-535
bool matches_pattern(const std::string &s, const std::string &pattern)
{
    if (pattern.empty()) return s.empty();
    size_t si = 0, pi = 0, match_idx = 0;
    long star_idx = -1;
    while (si < s.size()) {
        if (pi < pattern.size() && (pattern[pi] == '?' || pattern[pi] == s[si])) { si++; pi++; }
        else if (pi < pattern.size() && pattern[pi] == '*') { star_idx = (long)pi; match_idx = si; pi++; }
        else if (star_idx != -1) { pi = (size_t)star_idx + 1; match_idx++; si = match_idx; }
        else return false;
    }
    while (pi < pattern.size() && pattern[pi] == '*') pi++;
    return pi == pattern.size();
}

// cmd/main.v:482-500
bool should_include(const std::string &f, const std::vector<std::string> &only, const std::vector<std::string> &nots)
{
    for (const std::string &p : nots) if (matches_pattern(f, p)) return false;
    if (!only.empty()) {
        for (const std::string &p : only) if (matches_pattern(f, p)) return true;
        return false;
    }
    return true;
}

bool is_dir(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode); }
bool exists(const std::string &p) { struct stat st; return stat(p.c_str(), &st) == 0; }

// cmd/main.v:463-479
void collect_files(const std::string &dir, const Config &cfg, std::vector<std::string> *out)
{
    DIR *d = opendir(dir.c_str());
    if (!d) return;
    std::vector<std::string> names;
    while (dirent *e = readdir(d)) {
        const std::string nm = e->d_name;
        if (nm != "." && nm != "..") names.push_back(nm);
    }
    closedir(d);
    for (const std::string &nm : names) {
        const std::string path = dir + "/" + nm;
        if (is_dir(path)) collect_files(path, cfg, out);
        else if (should_include(path, cfg.only_files, cfg.not_files)) out->push_back(path);
    }
}

bool read_file(const std::string &p, std::vector<uint8_t> *v)
{
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    uint8_t buf[1 << 16];
    size_t k;
    v->clear();
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) v->insert(v->end(), buf, buf + k);
    fclose(f);
    return true;
}
bool write_file(const std::string &p, const uint8_t *d, size_t n)
{
    FILE *f = fopen(p.c_str(), "wb");
    if (!f) return false;
    const bool ok = n == 0 || fwrite(d, 1, n, f) == n;
    return fclose(f) == 0 && ok;
}
std::string base_name(const std::string &p) { const size_t i = p.find_last_of('/'); return i == std::string::npos ? p : p.substr(i + 1); }
std::string dir_name(const std::string &p) { const size_t i = p.find_last_of('/'); return i == std::string::npos ? "." : (i == 0 ? "/" : p.substr(0, i)); }
void mkdir_all(const std::string &p)
{
    std::string cur;
    for (size_t i = 0; i <= p.size(); i++) {
        if (i == p.size() || p[i] == '/') { if (!cur.empty() && !exists(cur)) mkdir(cur.c_str(), 0777); }
        if (i < p.size()) cur.push_back(p[i]);
    }
}

void print_usage()
{
    puts("zpaqv - ZPAQ archiver, MI355X batch coder behind the reference CLI");
    puts("Usage: zpaqv command archive[.zpaq] files... -options...");
    puts("Commands:");
    puts("  a, add      Append files to archive (one block per file, all blocks coded as one GPU batch)");
    puts("  x, extract  Extract files");
    puts("  l, list     List archive contents");
    puts("Options:");
    puts("  -f, -force      Add: overwrite archive instead of appending. Extract: overwrite files.");
    puts("  -mN, -method N  Compression level 0..5 (default 1).");
    puts("  -not files...   Exclude. * and ? match any string or char.");
    puts("  -only files...  Include only matches (default: *).");
    puts("  -sN, -summary N Brief progress.");
    puts("  -test           Extract: verify but do not write files.");
    puts("  -to out...      Extract: use out[0] as output directory prefix.");
    puts("  -tN, -threads N Spread the blocks over N GPUs (block b -> GPU b mod N; 0 = all; default 1).");
    puts("  -fragment N     Add: cut files into blocks of 2^N KiB (0..20) so that one big file becomes many");
    puts("                  independent blocks for the GPU.  Not reference behaviour (it ignores the option):");
    puts("                  without it an archive is byte-identical to the reference CLI's.");
}

bool parse_args(int argc, char **argv, Config *cfg, std::string *err)
{
    std::vector<std::string> rest;
    std::vector<std::string> *multi = nullptr;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a.size() >= 2 && a[0] == '-') {
            std::string name = a.substr(a[1] == '-' ? 2 : 1);
            multi = nullptr;
            // zpaq-style -mN -sN -tN (cmd/main.v:165-194)
            if (name.size() >= 2 && (name[0] == 'm' || name[0] == 's' || name[0] == 't') && is_numeric(name.substr(1))) {
                const int v = atoi(name.c_str() + 1);
                if (name[0] == 'm') cfg->method = v; else if (name[0] == 's') cfg->summary = v; else cfg->threads = v;
                continue;
            }
            auto need = [&](int *dst) -> bool {
                if (i + 1 >= argc) { *err = "Missing value for -" + name; return false; }
                const int v = atoi(argv[++i]);
                if (dst) *dst = v;
                return true;
            };
            if (name == "f" || name == "force") cfg->force = true;
            else if (name == "test") cfg->test_mode = true;
            else if (name == "noattributes") {}
            else if (name == "m" || name == "method") { if (!need(&cfg->method)) return false; }
            else if (name == "s" || name == "summary") { if (!need(&cfg->summary)) return false; }
            else if (name == "fragment") { if (!need(&cfg->fragment)) return false; }
            else if (name == "t" || name == "threads") { if (!need(&cfg->threads)) return false; }
            else if (name == "all" || name == "until") { if (!need(nullptr)) return false; }
            else if (name == "index" || name == "key" || name == "repack") { if (i + 1 < argc) i++; }
            else if (name == "not") multi = &cfg->not_files;
            else if (name == "only") multi = &cfg->only_files;
            else if (name == "to") multi = &cfg->to_files;
            else if (name == "h" || name == "help") { cfg->command = "help"; return true; }
            else { *err = "Unknown option '" + a + "'"; return false; }
            continue;
        }
        if (multi) multi->push_back(a); else rest.push_back(a);
    }
    if (rest.empty()) { *err = "Missing command. Use: add (a), extract (x), or list (l)"; return false; }
    std::string c = rest[0];
    for (char &ch : c) ch = (char)tolower(ch);
    if (c == "a" || c == "add") cfg->command = "add";
    else if (c == "x" || c == "extract") cfg->command = "extract";
    else if (c == "l" || c == "list") cfg->command = "list";
    else if (c == "help") { cfg->command = "help"; return true; }
    else if (c == "__match" || c == "__include") {          // test hooks for the reference's CLI known-answer tables (cmd/main_test.v)
        cfg->command = c;
        cfg->files.assign(rest.begin() + 1, rest.end());
        return true;
    }
    else { *err = "Unknown command '" + c + "'. Use: add (a), extract (x), or list (l)"; return false; }
    if (rest.size() < 2) { *err = "Missing archive name"; return false; }
    cfg->archive = rest[1];
    cfg->files.assign(rest.begin() + 2, rest.end());
    return true;
}

std::string archive_name(const Config &cfg)
{
    const std::string &a = cfg.archive;
    return (a.size() >= 5 && a.compare(a.size() - 5, 5, ".zpaq") == 0) ? a : a + ".zpaq";
}

// one context per GPU: devices ZPAQV_DEVICE, +1, ... (-tN; 0 = as many as there are).  {nullptr} if none works.
std::vector<zpq_ctx *> open_ctxs(const Config &cfg, bool required)
{
    std::vector<zpq_ctx *> ctxs;
    const char *dev = getenv("ZPAQV_DEVICE");
    const int first = dev ? atoi(dev) : 0;
    const int want = cfg.threads > 0 ? cfg.threads : 64;
    int rc = ZPQ_OK;
    for (int i = 0; i < want; i++) {
        zpq_ctx *c = nullptr;
        rc = zpq_ctx_create(first + i, &c);
        if (rc != ZPQ_OK) break;
        ctxs.push_back(c);
    }
    if (ctxs.empty()) {
        if (required) fprintf(stderr, "zpaqv: no usable MI355X device (%s); only -m0 archives can be handled\n", zpq_status_string(rc));
        ctxs.push_back(nullptr);
    }
    return ctxs;
}
void close_ctxs(std::vector<zpq_ctx *> &ctxs) { for (zpq_ctx *c : ctxs) if (c) zpq_ctx_destroy(c); ctxs.clear(); }

// cmd/main.v:239-327
int run_add(const Config &cfg)
{
    const std::string archive = archive_name(cfg);
    std::vector<std::string> to_add;
    for (const std::string &f : cfg.files) {
        if (is_dir(f)) collect_files(f, cfg, &to_add);
        else if (exists(f)) { if (should_include(f, cfg.only_files, cfg.not_files)) to_add.push_back(f); }
        else fprintf(stderr, "Warning: '%s' not found, skipping\n", f.c_str());
    }
    if (to_add.empty()) { fprintf(stderr, "No files to add\n"); return 1; }
    std::vector<uint8_t> out;
    if (exists(archive) && !cfg.force) {
        if (!read_file(archive, &out)) { fprintf(stderr, "Warning: Could not read existing archive '%s', creating new archive\n", archive.c_str()); out.clear(); }
    }
    std::vector<zpaq::ArchiveFile> files;
    std::vector<std::string> paths;
    for (const std::string &f : to_add) {
        zpaq::ArchiveFile af;
        if (!read_file(f, &af.data)) { fprintf(stderr, "Warning: Could not read '%s', skipping\n", f.c_str()); continue; }
        af.name = base_name(f);
        af.comment = std::to_string(af.data.size()) + " bytes";
        files.push_back(std::move(af));
        paths.push_back(f);
    }
    std::vector<zpq_ctx *> ctxs = cfg.method > 0 ? open_ctxs(cfg, true) : std::vector<zpq_ctx *>{nullptr};
    const size_t frag = (cfg.fragment >= 0 && cfg.fragment <= 20) ? ((size_t)1024 << cfg.fragment) : 0;
    const int rc = zpaq::archive_add(ctxs, cfg.method, files, &out, frag);
    close_ctxs(ctxs);
    if (rc != ZPQ_OK) { fprintf(stderr, "zpaqv: add failed: %s\n", zpq_status_string(rc)); return 1; }
    if (cfg.summary > 0) for (const std::string &p : paths) printf("Added: %s\n", p.c_str());
    if (!write_file(archive, out.data(), out.size())) { fprintf(stderr, "Could not write archive: %s\n", archive.c_str()); return 1; }
    printf("Created archive: %s\n", archive.c_str());
    printf("Files added: %zu\n", files.size());
    return 0;
}

int load_and_extract(const Config &cfg, bool want_data, std::vector<zpaq::ArchiveFile> *files, std::string *archive)
{
    *archive = archive_name(cfg);
    if (!exists(*archive)) { fprintf(stderr, "Archive '%s' not found\n", archive->c_str()); return 1; }
    std::vector<uint8_t> data;
    if (!read_file(*archive, &data)) { fprintf(stderr, "Could not read archive: %s\n", archive->c_str()); return 1; }
    std::vector<zpq_ctx *> ctxs = open_ctxs(cfg, false);
    // unnamed segments continue the previous file (archives written with -fragment); the reference CLI never writes them
    const int rc = zpaq::archive_extract(ctxs, data.data(), data.size(), want_data, files, true);
    close_ctxs(ctxs);
    if (rc != ZPQ_OK) { fprintf(stderr, "zpaqv: reading the archive failed: %s\n", zpq_status_string(rc)); return 1; }
    return 0;
}

// cmd/main.v:330-413
int run_extract(const Config &cfg)
{
    std::vector<zpaq::ArchiveFile> files;
    std::string archive;
    if (int r = load_and_extract(cfg, !cfg.test_mode, &files, &archive)) return r;
    int extracted = 0, bad = 0;
    for (const zpaq::ArchiveFile &f : files) {
        if (!should_include(f.name, cfg.only_files, cfg.not_files)) continue;
        const std::string out_name = cfg.to_files.empty() ? f.name : cfg.to_files[0] + "/" + f.name;
        if (f.status != ZPQ_OK) { fprintf(stderr, "Warning: '%s' could not be decoded: %s\n", f.name.c_str(), zpq_status_string(f.status)); bad++; continue; }
        if (!f.sha1_ok) { fprintf(stderr, "Warning: '%s' SHA-1 mismatch\n", f.name.c_str()); bad++; }
        if (!cfg.test_mode) {
            if (exists(out_name) && !cfg.force) { fprintf(stderr, "Warning: '%s' exists, skipping (use -force to overwrite)\n", out_name.c_str()); continue; }
            const std::string dir = dir_name(out_name);
            if (dir != "." && !exists(dir)) mkdir_all(dir);
            if (!write_file(out_name, f.data.data(), f.data.size())) { fprintf(stderr, "Warning: Could not write '%s'\n", out_name.c_str()); continue; }
        }
        extracted++;
        if (cfg.summary > 0 || cfg.test_mode) printf("%s: %s\n", cfg.test_mode ? "Verified" : "Extracted", out_name.c_str());
    }
    printf("Files %s: %d\n", cfg.test_mode ? "verified" : "extracted", extracted);
    return bad ? 2 : 0;
}

// cmd/main.v:422-470
int run_list(const Config &cfg)
{
    std::vector<zpaq::ArchiveFile> files;
    std::string archive;
    if (int r = load_and_extract(cfg, false, &files, &archive)) return r;
    printf("Contents of %s:\n", archive.c_str());
    puts("----------------------------------------");
    int total = 0;
    for (const zpaq::ArchiveFile &f : files) {
        if (!should_include(f.name, cfg.only_files, cfg.not_files)) continue;
        if (!f.comment.empty()) printf("%s (%s)\n", f.name.c_str(), f.comment.c_str());
        else printf("%s\n", f.name.c_str());
        total++;
    }
    puts("----------------------------------------");
    printf("Total files: %d\n", total);
    return 0;
}

}  // namespace

int main(int argc, char **argv)
{
    Config cfg;
    std::string err;
    if (!parse_args(argc, argv, &cfg, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    if (cfg.command == "help") { print_usage(); return 0; }
    if (cfg.command == "__match") {
        const bool m = cfg.files.size() == 2 ? matches_pattern(cfg.files[0], cfg.files[1]) : (cfg.files.size() == 1 ? matches_pattern("", cfg.files[0]) : false);
        puts(m ? "true" : "false");
        return 0;
    }
    if (cfg.command == "__include") {
        puts(!cfg.files.empty() && should_include(cfg.files[0], cfg.only_files, cfg.not_files) ? "true" : "false");
        return 0;
    }
    if (cfg.command == "add") return run_add(cfg);
    if (cfg.command == "extract") return run_extract(cfg);
    return run_list(cfg);
}
