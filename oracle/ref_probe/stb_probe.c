/* Probe of the reference's VENDORED texture decoder (stb_image.h under /root/reference/include, compiled where it lies and
 * as it is: plain C, no stand-ins).  For every file on the command line it prints what the reference's loader receives from
 * stbi_load(path, &x, &y, &comp, 0) (reference: include/Loader.h:58): x, y, comp and the samples (a hash and the first 24 on
 * stdout; all of them into $STB_PROBE_DUMP/<file>.raw when that variable names a directory).
 * Built only in the authoring container (oracle/Makefile target ref_probe); its output is committed as
 * tests/golden/stb_decode.json by tests/golden/make_texture_golden.py.  Nothing here is product code. */
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv)
{
    printf("{\n");
    for (int i = 1; i < argc; i++) {
        int x = 0, y = 0, comp = 0;
        unsigned char* p = stbi_load(argv[i], &x, &y, &comp, 0);
        const char* name = strrchr(argv[i], '/');
        name = name ? name + 1 : argv[i];
        if (!p) {
            printf("%s \"%s\": {\"error\": \"%s\"}", i > 1 ? ",\n" : "", name, stbi_failure_reason());
            continue;
        }
        unsigned long long h = 1469598103934665603ull; /* FNV-1a 64 over the samples */
        const size_t n = (size_t)x * y * comp;
        for (size_t k = 0; k < n; k++) { h ^= p[k]; h *= 1099511628211ull; }
        printf("%s \"%s\": {\"x\": %d, \"y\": %d, \"comp\": %d, \"fnv1a64\": \"%016llx\", \"head\": [", i > 1 ? ",\n" : "", name, x, y, comp, h);
        for (size_t k = 0; k < n && k < 24; k++) printf("%s%d", k ? "," : "", p[k]);
        printf("]}");
        if (getenv("STB_PROBE_DUMP")) { /* every sample, for the fixtures no second decoder reproduces (JPEG): <dir>/<name>.raw */
            char out[4096];
            snprintf(out, sizeof(out), "%s/%s.raw", getenv("STB_PROBE_DUMP"), name);
            FILE* f = fopen(out, "wb");
            if (f) { fwrite(p, 1, n, f); fclose(f); }
        }
        stbi_image_free(p);
    }
    printf("\n}\n");
    return 0;
}
