// jd_cli.cpp -- `jpegdec file.jpg ...`: the reference's CLI (main, main.go:1503-1513) on top of the C-ABI.
// Decodes each file in turn and writes <basename>.bmp into the current directory, byte-identical in layout
// to writeBitMap (main.go:1171-1218).  Errors print the reference-style message and exit 1.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/jpeghip.h"

static int fail(const char* what, int rc) {
    printf("%s: %s\n", what, jd_strerror(rc));
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 2) { printf("Error! No file given\n"); return 1; }
    printf("***** JPEG Decoder (MI355X hot path) *****\n\n");
    int rc = jd_init(nullptr, 0);
    if (rc) return fail("jd_init", rc);
    for (int a = 1; a < argc; a++) {
        FILE* f = fopen(argv[a], "rb");
        if (!f) { printf("Error! cannot open %s\n", argv[a]); return 1; }
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        std::vector<uint8_t> file((size_t)(n > 0 ? n : 0));
        if (fread(file.data(), 1, file.size(), f) != file.size()) { fclose(f); printf("Error! EOF\n"); return 1; }
        fclose(f);
        jd_frame fr; size_t off = 0, len = 0;
        rc = jd_parse(file.data(), file.size(), &fr, &off, &len);
        if (rc) return fail(argv[a], rc);
        printf("len(ecs) = %zu\n", len);
        std::vector<uint8_t> rgb((size_t)fr.width * fr.height * 3);
        rc = jd_decode(&fr, file.data() + off, len, 1, nullptr, 0, rgb.data(), (size_t)fr.width * 3, 0);
        if (rc) return fail(argv[a], rc);
        std::string name = argv[a];
        size_t slash = name.find_last_of('/'); if (slash != std::string::npos) name = name.substr(slash + 1);
        size_t dot = name.find_last_of('.'); if (dot != std::string::npos) name = name.substr(0, dot);
        name += ".bmp";
        long sz = jd_bmp_encode(nullptr, 0, fr.width, fr.height, nullptr, 0);
        std::vector<uint8_t> bmp((size_t)sz);
        jd_bmp_encode(rgb.data(), (size_t)fr.width * 3, fr.width, fr.height, bmp.data(), bmp.size());
        printf("Writing bitmap to %s ... \n", name.c_str());
        FILE* o = fopen(name.c_str(), "wb");
        if (!o || fwrite(bmp.data(), 1, bmp.size(), o) != bmp.size()) { printf("Error! cannot write %s\n", name.c_str()); return 1; }
        fclose(o);
    }
    jd_shutdown();
    return 0;
}
