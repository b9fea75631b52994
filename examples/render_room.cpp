/*
 * render_room.cpp - the reference's main() (raytracer.cpp:1153-1515) written against rt_host.hpp: the same
 * object definitions and calls, with the pixel loop replaced by rt::render(). Scene: light + S1/S2/S3 + walls +
 * floor (:1164-1178) and, if an OBJ path is given, one mesh with the rifle soldier's transform chain
 * (:1245-1251). Writes a PNG (name ending in .png) or a binary PPM.
 *   render_room <out.png|out.ppm> <W> <spp> [mesh.obj [texture.ppm]]
 */
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <vector>

#include "rt_host.hpp"

using namespace rt;

int main(int argc, char **argv) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s out.ppm W spp [mesh.obj [texture.ppm]]\n", argv[0]);
        return 2;
    }
    int W = atoi(argv[2]), H = W;
    try {
        Scene scene;
        scene.I = 6E9;
        scene.L = Vector(-10, 50, 40);
        Sphere Slum(scene.L, 5, Vector(1., 1., 1.));
        Sphere S1(Vector(-15, 0, 0), 10, Vector(0., 0., 1.));
        Sphere S2(Vector(0, 0, 0), 10, Vector(1., 1., 1.), true);
        Sphere S3(Vector(15, 0, 0), 10, Vector(1., 0., 0.), false, true);
        Sphere Smurga(Vector(-1050, 0, 0), 970, Vector(138 / 255., 168 / 255., 167 / 255.));
        Sphere Smurdr(Vector(1050, 0, 0), 970, Vector(138 / 255., 168 / 255., 167 / 255.));
        Sphere Smurfa(Vector(0, 0, -1050), 940, Vector(138 / 255., 168 / 255., 167 / 255.));
        Sphere Smurde(Vector(0, 0, 1000), 940, Vector(138 / 255., 168 / 255., 167 / 255.));
        Sphere Ssol(Vector(0, -1000, 0), 990, Vector(1., 1., 1.), false);
        TriangleMesh m(Vector(0., 0., 1.), false, false);
        scene.objects.push_back(&Slum);
        scene.objects.push_back(&S1);
        scene.objects.push_back(&S2);
        scene.objects.push_back(&S3);
        scene.objects.push_back(&Smurga);
        scene.objects.push_back(&Smurdr);
        scene.objects.push_back(&Smurfa);
        scene.objects.push_back(&Smurde);
        scene.objects.push_back(&Ssol);
        if (argc > 4) {
            m.readOBJ(argv[4]);
            if (argc > 5) m.loadTexture(argv[5]);
            m.zoom(4);
            m.rotate(1, 2);
            m.rotate(0, 2);
            m.translate(1, -10);
            m.rotateAroundAxe(1, -30);
            m.translate(0, 5);
            m.translate(2, 25);
            m.buildBVH();
            scene.objects.push_back(&m);
        }
        Camera cam; /* C = (0,30,55), fov 60, -25 degrees: the committed camera */
        RenderParams rp;
        rp.W = W;
        rp.H = H;
        rp.nbrays = atoi(argv[3]);
        std::vector<unsigned char> image;
        render(scene, cam, rp, image);
        const size_t len = strlen(argv[1]);
        if (len > 4 && strcmp(argv[1] + len - 4, ".png") == 0) { /* as main() ends: raytracer.cpp:1505-1512 */
            writePNG(argv[1], W, H, &image[0]);
        } else {
            FILE *f = fopen(argv[1], "wb");
            if (!f) return 3;
            fprintf(f, "P6\n%d %d\n255\n", W, H);
            fwrite(&image[0], 1, image.size(), f);
            fclose(f);
        }
    } catch (const std::exception &e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}
