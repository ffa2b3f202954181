// fuzz_readers.cpp -- tests/test_host.py builds this with -fsanitize=address,undefined: the image readers of the driver (png.h, tiff.h) must reject or
// decode byte-mutated and truncated files without touching memory they do not own.  usage: fuzz_readers <mutants per file> <tmp file> <file>...
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "png.h"
#include "tiff.h"

int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int per_file = atoi(argv[1]);
    const char *tmp = argv[2];
    unsigned seed = 1;
    int decoded = 0, total = 0;
    for (int a = 3; a < argc; a++) {
        FILE *f = fopen(argv[a], "rb");
        if (!f) return 3;
        std::vector<unsigned char> d;
        for (int c; (c = fgetc(f)) != EOF;) d.push_back((unsigned char)c);
        fclose(f);
        const bool tiff = strlen(argv[a]) > 4 && !strcmp(argv[a] + strlen(argv[a]) - 4, ".tif");
        for (int it = 0; it < per_file; it++) {
            std::vector<unsigned char> m = d;
            const int nm = 1 + rand_r(&seed) % 6;
            for (int k = 0; k < nm; k++) {
                size_t pos = (size_t)rand_r(&seed) % m.size();
                if (it % 3 == 0) pos = (size_t)(rand_r(&seed) % 300) % m.size();                 // headers and directories sit at the ends of the file
                if (it % 5 == 4) pos = m.size() - 1 - (size_t)(rand_r(&seed) % 200) % m.size();
                m[pos] = (unsigned char)rand_r(&seed);
            }
            if (it % 7 == 6) m.resize((size_t)rand_r(&seed) % m.size());
            FILE *o = fopen(tmp, "wb");
            if (!o) return 4;
            fwrite(m.data(), 1, m.size(), o);
            fclose(o);
            png_image img;
            if (tiff ? tiff_read(tmp, img) : png_read(tmp, img)) {
                decoded++;
                if (img.samples.size() != (size_t)img.width * img.height * img.channels) return 5;      // whatever is accepted is self-consistent
            }
            total++;
        }
    }
    printf("fuzz: %d of %d mutants decoded, none crashed\n", decoded, total);
    return 0;
}
