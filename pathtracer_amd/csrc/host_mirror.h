// C++ host-side mirror of the two reference classes the CLI talks to, implemented over the C ABI:
//   wa_content    (src/content.h:18-57)                  scene container, owns the GPU-resident scene
//   udp_renderer  (src/unidirectionalPathTracer.h:27-48)  constructor + render(float*)
// so that main() reads like the reference's (src/myRenderer.cpp:141-156).
#pragma once
#include <stdexcept>
#include <string>

#include "../../include/wapt.h"

struct wa_shader_tag {};  // globalShader travels inside the scene description

class wa_content {
  public:
    wa_content(const std::string& waff_path, int device = 0) {
        if (wapt_waff_load(waff_path.c_str(), &waff_)) throw std::runtime_error(wapt_last_error());
        wapt_waff_desc(waff_, &desc_);
        wapt_waff_settings(waff_, &td, &me, &mr);
        if (wapt_scene_create(&desc_, device, &scene_)) {
            std::string e = wapt_last_error();
            wapt_waff_free(waff_);
            throw std::runtime_error(e);
        }
    }
    ~wa_content() {
        wapt_scene_destroy(scene_);
        wapt_waff_free(waff_);
    }
    wa_content(const wa_content&) = delete;
    wa_content& operator=(const wa_content&) = delete;
    void getImageResolution(int* resI, int* resJ) const {
        *resI = desc_.camera.resI;
        *resJ = desc_.camera.resJ;
    }
    wapt_scene* handle() const { return scene_; }
    int32_t td = 0;       // traceDepth                           (src/myRenderer.cpp:17)
    float me = 0, mr = 0; // rayMarchDistanceDistributionExponent / SampleRate (:18-19)
  private:
    wapt_waff* waff_ = nullptr;
    wapt_scene* scene_ = nullptr;
    wapt_scene_desc desc_;
};

class udp_renderer {
  public:
    udp_renderer(wa_content* content, float* frameBuffer, int psamples, int hsamples, int sphereSamples, int lsamples,
                 wa_shader_tag, int maxTraceDepth, float rmdde, float rmdsr)
        : content_(content), fb_(frameBuffer) {
        (void)sphereSamples;  // unused by the reference as well (shadowed local, cpp:591)
        p_ = wapt_render_params{};
        p_.psamples = psamples;
        p_.hsamples = hsamples;
        p_.lsamples = lsamples;
        p_.max_depth = maxTraceDepth;
        p_.me = rmdde;
        p_.mr = rmdsr;
        p_.seed = 0;
        p_.sample_begin = 0;
        p_.sample_count = psamples;
    }
    ~udp_renderer() { wapt_progressive_destroy(prog_); }
    void setSeed(uint64_t s) { p_.seed = s; }
    // One call renders the whole image; once the progressive interface below has been touched it is one
    // iteration of the running average instead (cpp:141-161).
    void render(float* buffer) {
        fb_ = buffer;
        int rc = prog_ ? wapt_progressive_render(prog_, fb_, &stats) : wapt_render(content_->handle(), &p_, fb_, &stats);
        if (rc) throw std::runtime_error(wapt_last_error());
    }
    // progressive interface (src/unidirectionalPathTracer.h:40-47)
    void reset() { chk(wapt_progressive_reset(prog())); }
    void setIterationCount(int n) { chk(wapt_progressive_set_iteration(prog(), n)); }
    void setRenderResolution(int downresFactorOf2) { chk(wapt_progressive_set_resolution(prog(), downresFactorOf2)); }
    void setRegion(int l, int r, int b, int t) { chk(wapt_progressive_set_region(prog(), l, r, b, t)); }
    void abort() { chk(wapt_progressive_abort(prog())); }
    int generateTiles(int subdivLevel, wapt_tile* tiles) {
        int resI, resJ, n = 0;
        content_->getImageResolution(&resI, &resJ);
        chk(wapt_generate_tiles(resI, resJ, subdivLevel, tiles, &n));
        return n;
    }
    wapt_stats stats{};
  private:
    static void chk(int rc) {
        if (rc) throw std::runtime_error(wapt_last_error());
    }
    wapt_progressive* prog() {
        if (!prog_) chk(wapt_progressive_create(content_->handle(), &p_, &prog_));
        return prog_;
    }
    wa_content* content_;
    float* fb_;
    wapt_render_params p_;
    wapt_progressive* prog_ = nullptr;
};
