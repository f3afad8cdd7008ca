#!/bin/bash
# The host C code (scene / object / camera API, Nelder-Mead, kd builder, flatten, YAML) under AddressSanitizer + UBSan, CPU only
# (GPU sanitizers are not available on this pool): builds six scenes from the compiled reference's scene programs through the
# host API, flattens them, writes and re-reads them as YAML.  Needs oracle/_ref (build container).  Round 3: no finding in
# ndt_amd/host; the leaks LeakSanitizer reports are the reference scene programs' own (scenes/hypercube.c:37, 316-398).
set -e
cd "$(dirname "$0")/../.."
T=$(mktemp -d); cp -r ndt_amd/host/src ndt_amd/host/include $T/
sed -i "s|#include \"../../../include/ndt_hip.h\"|#include \"$PWD/include/ndt_hip.h\"|" $T/src/ndt_host_internal.h
cat > $T/drv.c <<'C'
#include "ndt_host_api.h"
#include "src/ndt_host_internal.h"
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int main(int argc, char **argv)
{
    register_objects(NULL);
    for (int a = 1; a + 1 < argc; a += 2) {
        void *h = dlopen(argv[a], RTLD_NOW | RTLD_GLOBAL);
        if (!h) { fprintf(stderr, "%s\n", dlerror()); return 2; }
        int (*setup)(scene *, int, int, int, char *) = (int (*)(scene *, int, int, int, char *))dlsym(h, "scene_setup");
        scene scn, back;
        memset(&scn, 0, sizeof(scn));
        setup(&scn, atoi(argv[a + 1]), 0, 1, NULL);
        ndt_flat_builder fb;
        char err[256];
        int rc = ndt_flatten_scene(&scn, &fb, err, sizeof(err));
        printf("%s -d %s: flatten rc %d, %d objects, %d kd nodes\n", argv[a], argv[a + 1], rc, fb.fs.n_objects, fb.fs.n_kd_nodes);
        ndt_flat_builder_free(&fb);
        scene_write_yaml(&scn, "/tmp/host_sanitize.yaml");
        memset(&back, 0, sizeof(back));
        scene_read_yaml(&back, "/tmp/host_sanitize.yaml", 0);
        scene_free(&back);
        scene_free(&scn);
    }
    return 0;
}
C
(cd $T && gcc -O1 -g -std=c99 -D_GNU_SOURCE -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -Iinclude -I. -w -o drv drv.c \
    src/ndt_vect.c src/ndt_nelder_mead.c src/ndt_bounding.c src/ndt_objects.c src/ndt_kdtree.c src/ndt_camera.c src/ndt_scene.c src/ndt_flatten.c src/ndt_yaml.c \
    -ldl -lm -lpthread -rdynamic)
S=oracle/_ref/scenes
ASAN_OPTIONS=detect_leaks=0 $T/drv $S/hypercube.so 3 $S/hypercube.so 6 $S/random.so 4 $S/balls.so 4 $S/parity_zoo.so 5 $S/parity_zoo.so 9 2>&1 | grep "flatten rc\|runtime error\|ERROR\|SUMMARY"
rm -rf $T
