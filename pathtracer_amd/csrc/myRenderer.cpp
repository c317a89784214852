// myRenderer — command line of the reference program (src/myRenderer.cpp:44-60, 80-198) on top of
// libwapt.so:  myRenderer -i scene.waff --psamples 64 --hsamples 1 --lsamples 1 -bmp image.bmp
// Options: --help/-h, --bmp (also -bmp, as the reference's README writes it), --input/-i (also the
// positional argument), --psamples, --hsamples, --lsamples, --convertOBJ --obj file.obj.
// Additions: --seed N, --device N.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "host_mirror.h"

static void usage() {
    std::cout << "Command line options:\n"
                 "  -h [ --help ]              Output this help message\n"
                 "  --bmp arg (=out.bmp)       Output BMP image\n"
                 "  -i [ --input ] arg (=in.waff) Input file\n"
                 "  --psamples arg (=1)        Number of samples per pixel\n"
                 "  --hsamples arg (=1)        Number of samples in the hemisphere\n"
                 "  --lsamples arg (=1)        Number of samples for each light\n"
                 "  --convertOBJ               Perform an OBJ conversion of a file, rather than a render\n"
                 "  --obj arg                  input OBJ file\n"
                 "  --seed arg (=0)            random seed\n"
                 "  --device arg (=0)          GPU index\n";
}

int main(int argc, char** argv) {
    std::string bmp = "out.bmp", input = "in.waff", obj;
    int psamples = 1, hsamples = 1, lsamples = 1, device = 0;
    unsigned long long seed = 0;
    bool convert = false, help = false;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto value = [&](const char* name) -> std::string {
            std::string eq = std::string(name) + "=";
            if (a.compare(0, eq.size(), eq) == 0) return a.substr(eq.size());
            if (i + 1 >= argc) {
                std::cerr << "the required argument for option '" << name << "' is missing\n";
                exit(EXIT_FAILURE);
            }
            return argv[++i];
        };
        auto is = [&](const char* name) { return a == name || a.compare(0, strlen(name) + 1, std::string(name) + "=") == 0; };
        if (a == "-h" || a == "--help") help = true;
        else if (is("--bmp") || is("-bmp")) bmp = value(a[1] == '-' ? "--bmp" : "-bmp");
        else if (is("--input") || a == "-i") input = value(a == "-i" ? "-i" : "--input");
        else if (is("--psamples")) psamples = atoi(value("--psamples").c_str());
        else if (is("--hsamples")) hsamples = atoi(value("--hsamples").c_str());
        else if (is("--lsamples")) lsamples = atoi(value("--lsamples").c_str());
        else if (is("--seed")) seed = strtoull(value("--seed").c_str(), nullptr, 10);
        else if (is("--device")) device = atoi(value("--device").c_str());
        else if (a == "--convertOBJ") convert = true;
        else if (is("--obj")) obj = value("--obj");
        else if (!a.empty() && a[0] == '-') {
            std::cerr << "unrecognised option '" << a << "'\n";
            return EXIT_FAILURE;
        } else input = a;  // positional = input
    }
    if (convert) {
        if (obj.size() < 4) {
            std::cerr << "--convertOBJ needs --obj file.obj\n";
            return EXIT_FAILURE;
        }
        std::string fast = obj.substr(0, obj.size() - 3) + "fast";
        std::cout << "Convert OBJ mode: " << obj << " -> " << fast << std::endl;
        if (wapt_convert_obj_to_fast(obj.c_str(), fast.c_str())) {
            std::cerr << wapt_last_error() << std::endl;
            return EXIT_FAILURE;
        }
        return 0;
    }
    if (help) {
        usage();
        return EXIT_SUCCESS;
    }
    try {
        wa_content content(input, device);
        int resI, resJ;
        content.getImageResolution(&resI, &resJ);
        std::vector<float> frameBuffer((size_t)resI * resJ * 3);
        std::cout << "Trace depth: " << content.td << "\nray march exp: " << content.me
                  << "\nray march rate: " << content.mr << std::endl;
        udp_renderer theRenderer(&content, frameBuffer.data(), psamples, hsamples, 1, lsamples, wa_shader_tag{},
                                 content.td, content.me, content.mr);
        theRenderer.setSeed(seed);
        auto t0 = std::chrono::steady_clock::now();
        theRenderer.render(frameBuffer.data());
        double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout << "Render time: " << sec << " s  (" << theRenderer.stats.paths / sec / 1e6 << " Mpaths/s, "
                  << theRenderer.stats.rays / sec / 1e6 << " Mrays/s)" << std::endl;
        wapt_tonemap(frameBuffer.data(), resI, resJ);
        if (wapt_write_bmp(bmp.c_str(), resI, resJ, frameBuffer.data())) {
            std::cerr << wapt_last_error() << std::endl;
            return EXIT_FAILURE;
        }
        // the reference also leaves an (empty) debug_content.obj behind on every run (src/myRenderer.cpp:192-195, 728-759)
        if (FILE* f = fopen("debug_content.obj", "w")) fclose(f);
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return EXIT_FAILURE;
    }
    return 0;
}
