// engine.hip -- the C-ABI of include/toybox_amd.h over the per-game gfx950 kernels.
// No CPU fallback: every entry point that computes needs a visible gfx950 device.

#include "tbx_common.hpp"
#include <emmintrin.h>   // host side only: tbx_host_stack_push

#include <sched.h>
#include <thread>
#include <vector>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

int tbx_agent_buffer(tbx_engine* e, int which, void** out_ptr, size_t* out_bytes);

namespace {

thread_local std::string g_create_error;

#define CHECK_ENGINE(e) \
    if (!(e)) return TBX_E_INVALID

}  // namespace

void tbx_set_create_error(const std::string& msg) { g_create_error = msg; }

void tbx_set_out_parity(tbx_engine* e, int p)
{
    e->out_par = p;
    e->reward = e->outs[p].reward;
    e->done = e->outs[p].done;
    e->lives_out = e->outs[p].lives;
    e->score_out = e->outs[p].score;
    e->packed = e->outs[p].packed;
    if (e->ops) e->ops->rebind_outputs(e);
}

namespace {

int hip_fail(tbx_engine* e, const char* what, hipError_t err)
{
    return e->fail(TBX_E_NO_DEVICE, std::string(what) + ": " + hipGetErrorString(err));
}

#define EHIP(call)                                        \
    do {                                                  \
        hipError_t _e = (call);                           \
        if (_e != hipSuccess) return hip_fail(e, #call, _e); \
    } while (0)

void breakout_default_config(tbx_breakout_config_t* c)
{
    memset(c, 0, sizeof *c);
    tbx_seed_state(13, c->rand[0], c->rand[1]);
    c->start_lives = 5;
    c->n_rows = 6;
    const int scores[6] = {7, 7, 4, 4, 1, 1};
    const uint8_t cols[6][3] = {{200, 72, 72}, {198, 108, 58}, {180, 122, 48}, {162, 162, 42}, {72, 160, 72}, {66, 72, 200}};
    for (int i = 0; i < 6; i++) {
        c->row_scores[i] = scores[i];
        c->row_colors[i] = tbx_color_t{cols[i][0], cols[i][1], cols[i][2], 255};
    }
    c->ball_speed_row_depth = 3;
    c->ball_speed_slow = 2.0;
    c->ball_speed_fast = 4.0;
    c->n_starts = 4;
    const double sx[4] = {24.0, 120.0, 120.0, 216.0}, sa[4] = {30.0, 30.0, 150.0, 150.0};
    for (int i = 0; i < 4; i++) {
        c->start_x[i] = sx[i]; c->start_y[i] = 80.0; c->start_angle_deg[i] = sa[i];
        const double rad = sa[i] * (M_PI / 180.0);
        c->start_dir_x[i] = std::cos(rad);   // host libm: the device never evaluates trig
        c->start_dir_y[i] = std::sin(rad);
    }
    c->paddle_discrete_segments = 5;
    for (int i = 0; i < 5; i++) {
        const double rad = (150.0 - (double)i * (120.0 / 4.0)) * (M_PI / 180.0);
        c->paddle_dir_x[i] = std::cos(rad);
        c->paddle_dir_y[i] = -std::sin(rad);
    }
    c->bg_color = tbx_color_t{0, 0, 0, 255};
    c->frame_color = tbx_color_t{144, 144, 144, 255};
    c->paddle_color = tbx_color_t{200, 72, 72, 255};
    c->ball_color = tbx_color_t{200, 72, 72, 255};
}

void si_default_config(tbx_si_config_t* c)
{
    memset(c, 0, sizeof *c);
    tbx_seed_state(17, c->rand[0], c->rand[1]);
    c->jitter = 0.5;
    c->start_lives = 3;
    c->n_rows = 6;
    c->n_shields = 3;
    c->enemy_protocol = 0;
    const int sc[6] = {30, 30, 20, 20, 10, 10}, sx[3] = {84, 148, 212};
    for (int i = 0; i < 6; i++) c->row_scores[i] = sc[i];
    for (int i = 0; i < 3; i++) { c->shield_x[i] = sx[i]; c->shield_y[i] = 157; }
}

// toybox/interventions/defaults/gridworld_config_default.json (tiles in the order of the dump's keys)
void gridworld_default_config(tbx_gridworld_config_t* c)
{
    static const char* const rows[7] = {"111111111", "1000R0001", "101111101", "100010001", "10001R111", "1000100G1", "111111111"};
    memset(c, 0, sizeof *c);
    c->width = 9; c->height = 7; c->n_tiles = 4;
    c->player_start_x = 2; c->player_start_y = 4;
    c->reward_becomes = 0;
    c->player_color = tbx_color_t{255, 0, 0, 255};
    const char keys[4] = {'0', '1', 'G', 'R'};
    const tbx_color_t colors[4] = {{255, 255, 255, 255}, {0, 0, 0, 255}, {0, 255, 0, 255}, {255, 255, 0, 255}};
    const int rewards[4] = {0, 0, 10, 1};
    for (int i = 0; i < 4; i++) {
        c->tile_keys[i] = (uint8_t)keys[i];
        c->tiles[i].color = colors[i];
        c->tiles[i].reward = rewards[i];
        c->tiles[i].goal = keys[i] == 'G';
        c->tiles[i].walkable = keys[i] != '1';
    }
    for (int y = 0; y < 7; y++)
        for (int x = 0; x < 9; x++)
            for (int i = 0; i < 4; i++)
                if (keys[i] == rows[y][x]) c->grid[y * TBX_GW_MAX_DIM + x] = (uint8_t)i;
}

void amidar_default_config(tbx_amidar_config_t* c)
{
    static const char* board[TBX_AMI_BOARD_H] = {
        "c========================c======", "=     =   =   =  =   =   =     =", "=     =   =   =  =   =   =     =",
        "=     =   =   =  =   =   =     =", "=     =   =   =  =   =   =     =", "=     =   =   =  =   =   =     =",
        "================================", "=   =    =  =      =  =    =   =", "=   =    =  =      =  =    =   =",
        "=   =    =  =      =  =    =   =", "=   =    =  =      =  =    =   =", "=   =    =  =      =  =    =   =",
        "================================", "=  =       =        =       =  p", "=  =       =        =       =  p",
        "=  =       =        =       =  p", "=  =       =        =       =  p", "=  =       =        =       =  p",
        "===============================p", "=    =        =  =        =    =", "=    =        =  =        =    =",
        "=    =        =  =        =    =", "=    =        =  =        =    =", "=    =        =  =        =    =",
        "c========================c======", "=     =     =      =     =     =", "=     =     =      =     =     =",
        "=     =     =      =     =     =", "=     =     =      =     =     =", "=     =     =      =     =     =",
        "================================"};
    memset(c, 0, sizeof *c);
    tbx_seed_state(13, c->rand[0], c->rand[1]);
    c->start_lives = 3; c->start_jumps = 4; c->jump_time = 75; c->chase_time = 300;
    c->box_bonus = 50; c->chase_score_bonus = 100;
    c->player_start_tx = 31; c->player_start_ty = 15;
    c->n_enemies = 5;
    c->render_images = 1; c->default_board_bugs = 1;
    for (int i = 0; i < 5; i++) {
        c->enemies[i].kind = TBX_AI_LOOKUP; c->enemies[i].default_route_index = i;
        c->enemies[i].seen_tx = c->enemies[i].seen_ty = -1;
    }
    c->bg_color = tbx_color_t{0, 0, 0, 255}; c->player_color = tbx_color_t{255, 255, 153, 255};
    c->unpainted_color = tbx_color_t{148, 0, 211, 255}; c->painted_color = tbx_color_t{255, 255, 30, 255};
    c->enemy_color = tbx_color_t{255, 50, 100, 255}; c->inner_painted_color = tbx_color_t{255, 255, 0, 255};
    for (int y = 0; y < TBX_AMI_BOARD_H; y++)
        for (int x = 0; x < TBX_AMI_BOARD_W; x++) {
            const char ch = board[y][x];
            c->board[y][x] = ch == '=' ? TBX_TILE_UNPAINTED : ch == 'p' ? TBX_TILE_PAINTED : ch == 'c' ? TBX_TILE_CHASE_MARKER : TBX_TILE_EMPTY;
        }
}

__global__ void seed_kernel(uint64_t* sim_rng, int n, int env, uint32_t seed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (env >= 0 && i != env) return;
    uint64_t s0, s1;
    tbx_seed_state(env >= 0 ? seed : seed + (uint32_t)i, s0, s1);
    sim_rng[i] = s0;
    sim_rng[(size_t)n + i] = s1;
}

__global__ void seed_array_kernel(uint64_t* sim_rng, int n, const uint32_t* seeds)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t s0, s1;
    tbx_seed_state(seeds[i], s0, s1);
    sim_rng[i] = s0;
    sim_rng[(size_t)n + i] = s1;
}

// step outputs into one block for the host: [reward N | lives N | score N | err 1 | done N bytes]
__global__ void gather_outputs_kernel(const int32_t* reward, const int32_t* lives, const int32_t* score, const uint8_t* done,
                                      uint32_t* err_flag, int32_t* out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { out[3 * (size_t)n] = (int32_t)*err_flag; *err_flag = 0u; }
    if (i >= n) return;
    out[i] = reward[i]; out[(size_t)n + i] = lives[i]; out[2 * (size_t)n + i] = score[i];
    reinterpret_cast<uint8_t*>(out + 3 * (size_t)n + 1)[i] = done[i];
}

__global__ void fill_rng_kernel(uint64_t* sim_rng, int n, uint64_t s0, uint64_t s1)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    sim_rng[i] = s0;
    sim_rng[(size_t)n + i] = s1;
}

int ensure_frame(tbx_engine* e, size_t bytes)
{
    if (e->frame_own_bytes < bytes) {
        if (e->frame_own) hipFree(e->frame_own);
        e->frame_own = nullptr;
        e->frame_own_bytes = 0;
        EHIP(hipMalloc((void**)&e->frame_own, bytes));
        e->frame_own_bytes = bytes;
    }
    e->frame = e->frame_own;
    e->frame_bytes = e->frame_own_bytes;
    return TBX_OK;
}

int check_err_flag(tbx_engine* e)
{
    uint32_t f = 0;
    EHIP(hipMemcpyAsync(&f, e->err_flag, sizeof f, hipMemcpyDeviceToHost, e->stream));
    EHIP(hipStreamSynchronize(e->stream));
    if (f) {
        EHIP(hipMemsetAsync(e->err_flag, 0, sizeof f, e->stream));
        if (f & 4u)
            return e->fail(TBX_E_NO_DEVICE, "an overlapped fused launch waited 3 s for the step blocks of the launch before it (TBX_OPT_FUSED_OVERLAP): results are not valid");
        if (f & 2u)
            return e->fail(TBX_E_NEEDS_RESET, "an env was stepped after its game ended inside EpisodicLifeEnv's no-op step (bench.Monitor raises here)");
        return e->fail(TBX_E_ACTION, "an illegal ALE action id was passed (treated as NOOP)");
    }
    return TBX_OK;
}

}  // namespace

extern "C" {

int tbx_abi_version(void) { return TBX_ABI_VERSION; }

const char* tbx_last_error(const tbx_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int tbx_frame_dims(int game, int* h, int* w)
{
    if (!h || !w) return TBX_E_INVALID;
    switch (game) {
    case TBX_GAME_BREAKOUT: *h = TBX_BRK_H; *w = TBX_BRK_W; return TBX_OK;
    case TBX_GAME_SPACE_INVADERS: *h = TBX_SI_H; *w = TBX_SI_W; return TBX_OK;
    case TBX_GAME_AMIDAR: *h = TBX_AMI_H; *w = TBX_AMI_W; return TBX_OK;
    case TBX_GAME_GRIDWORLD: *h = TBX_GW_H; *w = TBX_GW_W; return TBX_OK;
    default: return TBX_E_INVALID;
    }
}

int tbx_legal_actions(int game, int32_t* out, int cap)
{
    if (game < 0 || game >= TBX_NUM_GAMES) return TBX_E_INVALID;
    const int n = tbx_legal_count(game);
    for (int i = 0; i < n && i < cap; i++) out[i] = tbx_legal_action(game, i);
    return n;
}

int tbx_ale_action_to_buttons(int a)
{
    const uint32_t b = tbx_ale_buttons(a);
    return b == 0xFFu ? TBX_E_INVALID : (int)b;
}

size_t tbx_state_size(int game)
{
    switch (game) {
    case TBX_GAME_BREAKOUT: return sizeof(tbx_breakout_state_t);
    case TBX_GAME_SPACE_INVADERS: return sizeof(tbx_si_state_t);
    case TBX_GAME_AMIDAR: return sizeof(tbx_amidar_state_t);
    case TBX_GAME_GRIDWORLD: return sizeof(tbx_gridworld_state_t);
    default: return 0;
    }
}

size_t tbx_config_size(int game)
{
    switch (game) {
    case TBX_GAME_BREAKOUT: return sizeof(tbx_breakout_config_t);
    case TBX_GAME_SPACE_INVADERS: return sizeof(tbx_si_config_t);
    case TBX_GAME_AMIDAR: return sizeof(tbx_amidar_config_t);
    case TBX_GAME_GRIDWORLD: return sizeof(tbx_gridworld_config_t);
    default: return 0;
    }
}

int tbx_destroy(tbx_engine* e)
{
    if (!e) return TBX_OK;
    hipSetDevice(e->device);
    tbx_serve_stop(e);
    if (e->serve_stream) hipStreamDestroy(e->serve_stream);
    if (e->serve_ctl) hipHostFree(e->serve_ctl);
    if (e->serve_frame) hipHostFree(e->serve_frame);
    if (e->stream) hipStreamSynchronize(e->stream);
    TbxPipe& pp = e->pipe;
    for (int k = 0; k < 2; k++)
        if (pp.lane[k] && pp.lane[k] != e->stream) hipStreamSynchronize(pp.lane[k]);
    if (pp.step_lane) hipStreamSynchronize(pp.step_lane);
    tbx_gather_free(e);
    tbx_agent_free(e);
    if (e->ops) { e->ops->destroy(e); delete e->ops; }
    hipFree(e->sim_rng); hipFree(e->prev_score);
    for (int k = 0; k < 2; k++) {
        hipFree(e->outs[k].reward); hipFree(e->outs[k].done); hipFree(e->outs[k].lives); hipFree(e->outs[k].score); hipFree(e->outs[k].packed);
        hipFree(pp.frame[k]);
        if (pp.render_ev[k]) hipEventDestroy(pp.render_ev[k]);
        if (pp.user_step_ev[k]) hipEventDestroy(pp.user_step_ev[k]);
        if (pp.user_frame_ev[k]) hipEventDestroy(pp.user_frame_ev[k]);
        if (pp.launch_ev[k]) hipEventDestroy(pp.launch_ev[k]);
        if (pp.lane[k] && pp.lane[k] != e->stream) hipStreamDestroy(pp.lane[k]);
    }
    hipFree(pp.arrive);
    if (pp.step_lane) { hipStreamSynchronize(pp.step_lane); hipStreamDestroy(pp.step_lane); }
    for (int q = 0; q < 2; q++) {
        hipFree(pp.chunk_frames[q]); hipFree(pp.chunk_packed[q]);
        if (pp.chunk_step_ev[q]) hipEventDestroy(pp.chunk_step_ev[q]);
        for (int l = 0; l < 2; l++)
            if (pp.chunk_raster_ev[q][l]) hipEventDestroy(pp.chunk_raster_ev[q][l]);
    }
    hipFree(e->actions);
    hipFree(e->edit_args); hipFree(e->reduce_out);
    hipFree(e->mask); hipFree(e->err_flag); hipFree(e->frame_own); hipFree(e->staging); hipFree(e->scal); hipFree(e->one_frame); hipFree(e->io_dev);
    if (e->io_host) hipHostFree(e->io_host);
    if (e->scal_host) hipHostFree(e->scal_host);
    if (e->order_ev) hipEventDestroy(e->order_ev);
    if (pp.step_ev) hipEventDestroy(pp.step_ev);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
    return TBX_OK;
}

int tbx_device_identity(tbx_engine* e, tbx_device_identity_t* out)
{
    CHECK_ENGINE(e);
    if (!out) return e->fail(TBX_E_INVALID, "out is NULL");
    hipDeviceProp_t prop;
    EHIP(hipGetDeviceProperties(&prop, e->device));
    memset(out, 0, sizeof *out);
    out->ordinal = e->device;
    out->pci_domain = prop.pciDomainID; out->pci_bus = prop.pciBusID; out->pci_device = prop.pciDeviceID;
    out->total_memory = (uint64_t)prop.totalGlobalMem;
    out->compute_units = prop.multiProcessorCount;
    snprintf(out->arch, sizeof out->arch, "%s", prop.gcnArchName);
    snprintf(out->name, sizeof out->name, "%s", prop.name);
    return TBX_OK;
}

int tbx_create(int game, int n_envs, int device, const void* config_pod, size_t config_size, tbx_engine** out)
{
    if (!out) return TBX_E_INVALID;
    *out = nullptr;
    if (n_envs < 1) { g_create_error = "n_envs must be >= 1"; return TBX_E_INVALID; }
    int ndev = 0;
    hipError_t he = hipGetDeviceCount(&ndev);
    if (he != hipSuccess || ndev < 1) {
        g_create_error = std::string("no HIP device visible (the engine has no CPU fallback): ") + hipGetErrorString(he);
        return TBX_E_NO_DEVICE;
    }
    if (device < 0 || device >= ndev) { g_create_error = "device index out of range"; return TBX_E_INVALID; }
    hipDeviceProp_t prop;
    he = hipGetDeviceProperties(&prop, device);
    if (he != hipSuccess) { g_create_error = hipGetErrorString(he); return TBX_E_NO_DEVICE; }
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName + ", this library is built for gfx950 only";
        return TBX_E_NO_DEVICE;
    }
    tbx_engine* e = new (std::nothrow) tbx_engine();
    if (!e) { g_create_error = "out of host memory"; return TBX_E_NOMEM; }
    e->game = game;
    e->n = n_envs;
    e->device = device;
    int rc = TBX_OK;
    auto bail = [&](int code) {
        g_create_error = e->err;
        tbx_destroy(e);
        return code;
    };
    switch (game) {
    case TBX_GAME_BREAKOUT: e->ops = tbx_make_breakout_ops(); break;
    case TBX_GAME_SPACE_INVADERS: e->ops = tbx_make_si_ops(); break;
    case TBX_GAME_AMIDAR: e->ops = tbx_make_amidar_ops(); break;
    case TBX_GAME_GRIDWORLD: e->ops = tbx_make_gridworld_ops(); break;
    default: e->err = "unknown game id"; return bail(TBX_E_INVALID);
    }
    const size_t N = (size_t)n_envs;
#define CHIP(call)                                                                 \
    do {                                                                           \
        hipError_t _e = (call);                                                    \
        if (_e != hipSuccess) { hip_fail(e, #call, _e); return bail(_e == hipErrorOutOfMemory ? TBX_E_NOMEM : TBX_E_NO_DEVICE); } \
    } while (0)
    CHIP(hipSetDevice(device));
    CHIP(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    CHIP(hipMalloc((void**)&e->sim_rng, 2 * N * sizeof(uint64_t)));
    CHIP(hipMalloc((void**)&e->prev_score, N * sizeof(int32_t)));
    CHIP(hipMalloc((void**)&e->outs[0].reward, N * sizeof(int32_t)));
    CHIP(hipMalloc((void**)&e->outs[0].done, N));
    CHIP(hipMalloc((void**)&e->outs[0].lives, N * sizeof(int32_t)));
    CHIP(hipMalloc((void**)&e->outs[0].score, N * sizeof(int32_t)));
    CHIP(hipMalloc((void**)&e->outs[0].packed, N * sizeof(uint64_t)));
    tbx_set_out_parity(e, 0);
    CHIP(hipMalloc((void**)&e->actions, N * sizeof(int32_t)));
    CHIP(hipMalloc((void**)&e->mask, N));
    CHIP(hipMalloc((void**)&e->err_flag, sizeof(uint32_t)));
    CHIP(hipMalloc((void**)&e->scal, 3 * N * sizeof(int32_t)));
    CHIP(hipMemsetAsync(e->err_flag, 0, sizeof(uint32_t), e->stream));
    CHIP(hipMemsetAsync(e->reward, 0, N * sizeof(int32_t), e->stream));
    CHIP(hipMemsetAsync(e->done, 0, N, e->stream));
    CHIP(hipMemsetAsync(e->packed, 0, N * sizeof(uint64_t), e->stream));
    e->staging_bytes = e->ops->state_size();
    CHIP(hipMalloc(&e->staging, e->staging_bytes));

    // config: NULL -> game defaults
    std::vector<uint8_t> cfg(e->ops->config_size());
    if (config_pod) {
        if (config_size != cfg.size()) { e->err = "config size mismatch"; return bail(TBX_E_INVALID); }
        memcpy(cfg.data(), config_pod, cfg.size());
    } else {
        switch (game) {
        case TBX_GAME_BREAKOUT: breakout_default_config((tbx_breakout_config_t*)cfg.data()); break;
        case TBX_GAME_SPACE_INVADERS: si_default_config((tbx_si_config_t*)cfg.data()); break;
        case TBX_GAME_AMIDAR: amidar_default_config((tbx_amidar_config_t*)cfg.data()); break;
        case TBX_GAME_GRIDWORLD: gridworld_default_config((tbx_gridworld_config_t*)cfg.data()); break;
        }
    }
    rc = e->ops->init(e, cfg.data(), cfg.size());
    if (rc) return bail(rc);
    // every env's simulator RNG starts at config.rand (first 16 bytes of every config record)
    uint64_t r[2];
    memcpy(r, cfg.data(), sizeof r);
    hipLaunchKernelGGL(fill_rng_kernel, dim3((n_envs + 255) / 256), dim3(256), 0, e->stream, e->sim_rng, n_envs, r[0], r[1]);
    CHIP(hipGetLastError());
    rc = e->ops->new_game(e, nullptr, e->stream);
    if (rc) return bail(rc);
    CHIP(hipStreamSynchronize(e->stream));
#undef CHIP
    *out = e;
    return TBX_OK;
}

int tbx_num_envs(const tbx_engine* e) { return e ? e->n : TBX_E_INVALID; }
int tbx_game(const tbx_engine* e) { return e ? e->game : TBX_E_INVALID; }

int tbx_seed(tbx_engine* e, int env, uint32_t seed)
{
    CHECK_ENGINE(e);
    if (env < -1 || env >= e->n) return e->fail(TBX_E_INVALID, "env index out of range");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    hipLaunchKernelGGL(seed_kernel, dim3((e->n + 255) / 256), dim3(256), 0, e->stream, e->sim_rng, e->n, env, seed);
    EHIP(hipGetLastError());
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_seed_array(tbx_engine* e, const uint32_t* seeds_host)
{
    CHECK_ENGINE(e);
    if (!seeds_host) return e->fail(TBX_E_INVALID, "seeds pointer is NULL");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    // e->actions is the engine's N-dword upload staging
    EHIP(hipMemcpyAsync(e->actions, seeds_host, (size_t)e->n * sizeof(uint32_t), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(seed_array_kernel, dim3((e->n + 255) / 256), dim3(256), 0, e->stream, e->sim_rng, e->n,
                       reinterpret_cast<const uint32_t*>(e->actions));
    EHIP(hipGetLastError());
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_get_sim_rng(tbx_engine* e, int env, uint64_t out[2])
{
    CHECK_ENGINE(e);
    if (env < 0 || env >= e->n || !out) return e->fail(TBX_E_INVALID, "env index out of range");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    EHIP(hipMemcpyAsync(&out[0], e->sim_rng + env, 8, hipMemcpyDeviceToHost, e->stream));
    EHIP(hipMemcpyAsync(&out[1], e->sim_rng + (size_t)e->n + env, 8, hipMemcpyDeviceToHost, e->stream));
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_set_sim_rng(tbx_engine* e, int env, const uint64_t st[2])
{
    CHECK_ENGINE(e);
    if (env < -1 || env >= e->n || !st) return e->fail(TBX_E_INVALID, "env index out of range");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    if (env == -1) {
        hipLaunchKernelGGL(fill_rng_kernel, dim3((e->n + 255) / 256), dim3(256), 0, e->stream, e->sim_rng, e->n, st[0], st[1]);
        EHIP(hipGetLastError());
    } else {
        EHIP(hipMemcpyAsync(e->sim_rng + env, &st[0], 8, hipMemcpyHostToDevice, e->stream));
        EHIP(hipMemcpyAsync(e->sim_rng + (size_t)e->n + env, &st[1], 8, hipMemcpyHostToDevice, e->stream));
    }
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_new_game(tbx_engine* e, const uint8_t* mask_host)
{
    CHECK_ENGINE(e);
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    const uint8_t* m = nullptr;
    if (mask_host) {
        EHIP(hipMemcpyAsync(e->mask, mask_host, (size_t)e->n, hipMemcpyHostToDevice, e->stream));
        m = e->mask;
    }
    int rc = e->ops->new_game(e, m, e->stream);
    if (rc) return rc;
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_step_device(tbx_engine* e, const int32_t* actions_dev, uint32_t flags, void* stream)
{
    CHECK_ENGINE(e);
    if (!actions_dev) return e->fail(TBX_E_INVALID, "actions pointer is NULL");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, (hipStream_t)stream));
    EHIP(tbx_gather_before_step(e, (hipStream_t)stream));
    ActionSource src{};
    src.actions = actions_dev;
    src.single_env = -1;
    return e->ops->step(e, src, flags, (hipStream_t)stream);
}

// ---- pipelined mode (TBX_OPT_PIPELINE; contract in include/toybox_amd.h)
//
// A step with in-kernel actions depends on nothing the previous frame's rasteriser produces, and for games whose rasteriser
// reads step-written records (GameOps::pipeline_ok) it disturbs nothing that rasteriser reads once there are two buffers of
// records and of step outputs: it runs on the engine's step stream BESIDE the render queued before it.  Measured on MI355X
// (scripts/interleave_probe.py): a 10 us step kernel serialised between two 1.19 ms Breakout render launches costs
// 0.04-0.22 ms depending on the box, run beside the render it costs 0.01-0.04 ms.  With value 3 consecutive renders into the
// engine-owned frame buffer alternate between two internal streams and two buffers as well, so that launch N+1 ramps up in
// the ramp-down of launch N.
//
// Streams.  Value 2: one internal step stream; renders stay on the caller's stream U.  Value 3: two internal streams S[0],
// S[1]; step N and the overlapped render N both go to S[p], p = the parity of the buffers step N writes -- so render N is
// behind its step, behind render N-2 (same frame buffer) and behind step N-2 / render N-2's reads of records p by stream order
// alone, and runs beside render N-1 on the other stream.  (Few streams on purpose: the runtime multiplexes streams onto a
// handful of hardware queues -- GPU_MAX_HW_QUEUES, 4 by default -- and two streams that share one queue do not overlap.)
//
// Who waits for whom beyond stream order (U = the stream the caller names):
//   step N   (writes records / outputs p)   <- step N-1 (the state; other stream in value 3), a render of records p that ran
//                                              elsewhere, U's readers of outputs p (fence recorded on U at step N-1's call),
//                                              the gather that read outputs p
//   render N on U (value 2, or out_dev given) <- nothing: U waits for every step
//   render N on S[p] (value 3)              <- U's readers of frame buffer p (fence recorded on U at render N-1's call)
//   U                                        <- every step and every overlapped render (so whatever the caller queues next
//                                              sees them, and U is the tail that calls of any other kind join)
static int pipe_mode(const tbx_engine* e)
{
    const int v = e->opt[TBX_OPT_PIPELINE];
    if (v == 0 || e->gather_ring || !e->ops->pipeline_ok()) return 0;      // (the K-step record ring moves the one record pointer)
    if (v == 1) return e->ops->pipeline_auto(e->n, e->gather != nullptr);
    return v;
}

static int pipe_prepare(tbx_engine* e)
{
    TbxPipe& p = e->pipe;
    if (p.prepared) return TBX_OK;
    // every resource is made only while it is still missing, so that a call that failed half way can be repeated without
    // leaking what the first attempt got; `prepared` is set last
    const size_t N = (size_t)e->n;
    TbxStepOut& o = e->outs[1];
    if (!o.reward) { EHIP(hipMalloc((void**)&o.reward, N * sizeof(int32_t))); EHIP(hipMemset(o.reward, 0, N * sizeof(int32_t))); }
    if (!o.done) { EHIP(hipMalloc((void**)&o.done, N)); EHIP(hipMemset(o.done, 0, N)); }
    if (!o.lives) { EHIP(hipMalloc((void**)&o.lives, N * sizeof(int32_t))); EHIP(hipMemset(o.lives, 0, N * sizeof(int32_t))); }
    if (!o.score) { EHIP(hipMalloc((void**)&o.score, N * sizeof(int32_t))); EHIP(hipMemset(o.score, 0, N * sizeof(int32_t))); }
    if (!o.packed) { EHIP(hipMalloc((void**)&o.packed, N * sizeof(uint64_t))); EHIP(hipMemset(o.packed, 0, N * sizeof(uint64_t))); }
    if (!p.step_ev) EHIP(hipEventCreateWithFlags(&p.step_ev, hipEventDisableTiming));
    for (int k = 0; k < 2; k++) {
        if (!p.render_ev[k]) EHIP(hipEventCreateWithFlags(&p.render_ev[k], hipEventDisableTiming));
        if (!p.user_step_ev[k]) EHIP(hipEventCreateWithFlags(&p.user_step_ev[k], hipEventDisableTiming));
        if (!p.user_frame_ev[k]) EHIP(hipEventCreateWithFlags(&p.user_frame_ev[k], hipEventDisableTiming));
        if (!p.launch_ev[k]) EHIP(hipEventCreateWithFlags(&p.launch_ev[k], hipEventDisableTiming));
    }
    for (int q = 0; q < 2; q++) {
        if (!p.chunk_step_ev[q]) EHIP(hipEventCreateWithFlags(&p.chunk_step_ev[q], hipEventDisableTiming));
        for (int l = 0; l < 2; l++)
            if (!p.chunk_raster_ev[q][l]) EHIP(hipEventCreateWithFlags(&p.chunk_raster_ev[q][l], hipEventDisableTiming));
    }
    if (!p.arrive) {
        EHIP(hipMalloc((void**)&p.arrive, 2 * sizeof(unsigned long long)));
        EHIP(hipMemset(p.arrive, 0, 2 * sizeof(unsigned long long)));
        p.arrive_want = p.release_want = 0;
    }
    // The two internal streams are created with the highest priority.  Not for the priority's sake: the runtime multiplexes the
    // streams of a process onto a few hardware queues PER PRIORITY LEVEL (GPU_MAX_HW_QUEUES, 4 by default), two streams that
    // land on one queue do not overlap, and which streams share depends on the creation history of the whole process.  The
    // high-priority pool is the lanes' own (measured: with ordinary streams the overlapped modes were faster or slower than
    // the serial loop from one process to the next; scripts/pipeline_sweep.py).
    int lo = 0, hi = 0;
    EHIP(hipDeviceGetStreamPriorityRange(&lo, &hi));     // numerically hi <= lo
#ifdef TBX_DIAG
    if (getenv("TBX_LANE_PRIORITY")) hi = atoi(getenv("TBX_LANE_PRIORITY")) ? lo : hi;      // measurement builds: ordinary streams
#endif
    for (int k = 0; k < 2; k++)
        if (!p.lane[k]) EHIP(hipStreamCreateWithPriority(&p.lane[k], hipStreamNonBlocking, hi));
    if (!p.step_lane) EHIP(hipStreamCreateWithPriority(&p.step_lane, hipStreamNonBlocking, hi));   // (three of the priority level's four hardware queues)
    p.prepared = true;
    return TBX_OK;
}

// the first pipelined call after a call of any other kind: every internal stream behind all that came before
enum { PIPE_STEPS_AND_RENDERS = 0, PIPE_FUSED_OVERLAP = 1, PIPE_ROLLOUT_CHUNKS = 2 };

static int pipe_enter(tbx_engine* e, int kind = PIPE_STEPS_AND_RENDERS)
{
    TbxPipe& p = e->pipe;
    const bool fused = kind == PIPE_FUSED_OVERLAP, rollout = kind == PIPE_ROLLOUT_CHUNKS;
    if (e->pending_kind) EHIP(tbx_finish_pending(e));           // (a host-delivery step between its begin and end calls)
    // pipelined steps / renders, overlapped fused launches and rollout chunks keep different books on the same lanes: a change of
    // kind joins first (tbx_use_stream makes the stream of the last call wait for every internal launch; the lanes re-enter behind it)
    if (p.active && (p.fused != fused || p.rollout != rollout)) EHIP(tbx_use_stream(e, e->last_stream));
    if (p.active) return TBX_OK;
    if (e->serve_running) EHIP(tbx_serve_stop(e));
    int rc = pipe_prepare(e);
    if (rc) return rc;
    EHIP(tbx_wait_tail(e, p.lane[0]));
    EHIP(tbx_wait_tail(e, p.lane[1]));
    if (rollout) EHIP(tbx_wait_tail(e, p.step_lane));
    for (int k = 0; k < 2; k++) { p.render_pending[k] = p.user_step_rec[k] = p.user_frame_rec[k] = false; p.render_on[k] = nullptr; }
    p.step_outstanding = false;
    p.step_on = nullptr;
    p.step_user = nullptr;
    p.frame_par = -1;
    p.live_reader = -1;
    p.fused = fused;
    p.rollout = rollout;
    p.prev_overlapped = false;
    p.launch_rec[0] = p.launch_rec[1] = false;
    p.user_waits[0] = p.user_waits[1] = false;
    p.reader_seen = false;
    for (int q = 0; q < 2; q++) { p.chunk_step_rec[q] = p.chunk_raster_rec[q][0] = p.chunk_raster_rec[q][1] = p.chunk_user_waits[q] = false; }
    p.active = true;
    return TBX_OK;
}

static int pipe_step(tbx_engine* e, const ActionSource& src, uint32_t flags, hipStream_t user, int mode)
{
    int rc = pipe_enter(e);
    if (rc) return rc;
    TbxPipe& p = e->pipe;
    const int cur = e->out_par, wp = cur ^ 1;                  // this step writes output set wp ...
    const int rw = e->ops->records_parity() ^ 1;               // ... and records buffer rw
    hipStream_t ss = mode == 3 ? p.lane[wp] : p.lane[0];
    if (p.step_outstanding && p.step_on != ss) EHIP(hipStreamWaitEvent(ss, p.step_ev, 0));          // the state step N-1 left
    if (p.render_pending[rw]) {                                                                      // the reader of records rw
        if (p.render_on[rw] != ss) EHIP(hipStreamWaitEvent(ss, p.render_ev[rw], 0));
        p.render_pending[rw] = false;
    }
    if (p.user_step_rec[wp]) { EHIP(hipStreamWaitEvent(ss, p.user_step_ev[wp], 0)); p.user_step_rec[wp] = false; }
    // A render issued while the records did not reflect the state (after new_game / set_state / a single-env step) read LIVE
    // state -- the prep kernel that rebuilds the records, or a state-reading rasteriser -- on its own stream: this step, which
    // rewrites that state, goes behind it (it only waited for the reader of the OTHER records buffer).
    if (p.live_reader >= 0) {
        if (p.render_on[p.live_reader] != ss) EHIP(hipStreamWaitEvent(ss, p.render_ev[p.live_reader], 0));
        p.live_reader = -1;
    }
    // whatever the caller has queued so far may read the current outputs: the step after this one waits for it
    EHIP(hipEventRecord(p.user_step_ev[cur], user));
    p.user_step_rec[cur] = true;
    tbx_set_out_parity(e, wp);
    hipError_t ge = tbx_gather_before_step(e, ss);
    rc = ge == hipSuccess ? e->ops->step_ahead(e, src, flags, ss) : hip_fail(e, "tbx_gather_before_step", ge);
    if (rc) {                                                  // nothing was launched: TBX_BUF_* keep naming the set that holds results
        tbx_set_out_parity(e, cur);
        return rc;
    }
    EHIP(hipEventRecord(p.step_ev, ss));
    EHIP(hipStreamWaitEvent(user, p.step_ev, 0));
    p.step_outstanding = true;
    p.step_on = ss;
    p.step_user = user;
    e->last_stream = user;
    e->has_last = true;
    return TBX_OK;
}

static int pipe_render(tbx_engine* e, uint8_t* out_dev, int channels, hipStream_t user, int mode)
{
    int rc = pipe_enter(e);
    if (rc) return rc;
    TbxPipe& p = e->pipe;
    const int rp = e->ops->records_parity();
    const bool overlap = mode == 3 && out_dev == nullptr;
    const size_t bytes = (size_t)e->n * e->ops->height() * e->ops->width() * channels;
    hipStream_t rs = user;
    if (overlap) {
        const int fp = rp;                                     // frame buffer and stream follow the records' parity
        rs = p.lane[fp];
        if (p.frame_bytes[fp] < bytes) {
            EHIP(hipStreamSynchronize(rs));
            if (p.frame[fp]) hipFree(p.frame[fp]);
            p.frame[fp] = nullptr;
            p.frame_bytes[fp] = 0;
            EHIP(hipMalloc((void**)&p.frame[fp], bytes));
            p.frame_bytes[fp] = bytes;
        }
        if (p.step_outstanding && p.step_on != rs) EHIP(hipStreamWaitEvent(rs, p.step_ev, 0));
        // Readers of a frame are queued on U before the next render call.  Fence what is there now; the render into the OTHER
        // buffer waits for the fence of the call before (readers of what that buffer held), a second render into the SAME
        // buffer for the one just recorded.
        const int prev = p.frame_par < 0 ? fp ^ 1 : p.frame_par;
        EHIP(hipEventRecord(p.user_frame_ev[prev], user));
        p.user_frame_rec[prev] = true;
        if (p.user_frame_rec[fp]) { EHIP(hipStreamWaitEvent(rs, p.user_frame_ev[fp], 0)); p.user_frame_rec[fp] = false; }
        out_dev = p.frame[fp];
    } else {
        // on the caller's stream: it waits for the step unless it is the stream the step call named (pipe_step made that one wait)
        if (p.step_outstanding && p.step_user != user) EHIP(hipStreamWaitEvent(rs, p.step_ev, 0));
        if (!out_dev) {
            rc = ensure_frame(e, bytes);
            if (rc) return rc;
            out_dev = e->frame_own;
        }
    }
    if (((uintptr_t)out_dev & 15u) != 0) return e->fail(TBX_E_INVALID, "frame buffer must be 16-byte aligned");
    if (p.render_pending[rp] && p.render_on[rp] != rs) EHIP(hipStreamWaitEvent(rs, p.render_ev[rp], 0));   // (another render of these records)
    const bool reads_live = !e->ops->records_valid();          // the launch below starts from live state (pipe_step waits for it)
    rc = e->ops->render(e, out_dev, channels, 0, e->n, rs);
    if (rc) return rc;
    EHIP(hipEventRecord(p.render_ev[rp], rs));
    p.render_pending[rp] = true;
    p.render_on[rp] = rs;
    if (reads_live) p.live_reader = rp;
    if (overlap) {
        EHIP(hipStreamWaitEvent(user, p.render_ev[rp], 0));
        p.frame_par = rp;
        e->frame = p.frame[rp];
        e->frame_bytes = p.frame_bytes[rp];
    } else if (out_dev == e->frame_own) {
        e->frame = e->frame_own;
        e->frame_bytes = e->frame_own_bytes;
    }
    e->last_stream = user;
    e->has_last = true;
    return TBX_OK;
}

// ---- overlapped fused launches (TBX_OPT_FUSED_OVERLAP; contract in include/toybox_amd.h)
//
// Launch N of a loop of tbx_render_step_synthetic calls = [step blocks: state t -> t+1, records R[b], outputs O[wp]] +
// [rasteriser blocks: R[a] -> frame F[wp]].  Launch N+1 needs the step blocks of launch N and nothing else of it, so it goes to
// the OTHER lane (a stream per parity, like value 3 of the pipelined mode) behind a one-wave kernel that waits until the device
// counter `arrive` says those blocks are through (they fence and bump it last).  The launch itself never waits: a grid that
// spins could fill the chip before the launch it waits for has been dispatched.  Hazards and who orders them:
//   state, O[wp] written by step N, read / rewritten by step N+1     the counter (+ an acquire fence at the head of every wave)
//   R[b] written by step N, read by rasteriser N+1                   the counter
//   R[a] read by rasteriser N, rewritten by step N+2 (three buffers) stream order: N and N+2 share a lane
//   F[wp], O[wp] written by N, rewritten by N+2                       stream order; their READERS on the caller's stream U by the
//                                                                    fence recorded on U at call N+1 (user_step_ev), awaited by N+2
//   the collective that reads O[wp] / a ring                          tbx_gather waits for both lanes' completion events;
//                                                                    tbx_gather_before_step makes the rewriting launch wait for it
//   U                                                                 waits for every launch's completion event (what the caller
//                                                                    queues next sees the results; calls of other kinds join through U)
// Measured (scripts/ubench/overlap_ticket.hip, a stand-in launch, ms per launch serial / completion-event dependency / this /
// hipStreamWaitValue64 on the counter): 4 096 envs 0.0920 / 0.0982 / 0.0887 / 0.0886, 8 192: 0.1854 / 0.1917 / 0.1794 / 0.1791,
// 65 536: 1.555 / 1.560 / 1.477 / 1.483 -- the wait kernel costs nothing against the command processor's own (beta) wait.
__global__ void tbx_ticket_wait_kernel(const unsigned long long* arrive, unsigned long long want_steps, unsigned long long want_releases, uint32_t* err_flag)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    // (relaxed: what has to be seen behind the counter is read with agent-scope loads by the launch that follows)
    while (__hip_atomic_load(arrive + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want_releases ||
           __hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want_steps) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 300000000ull) { atomicOr(err_flag, 4u); return; }   // 3 s of the 100 MHz clock: report, never hang
    }
}

// the caller asks where a result of the last call lies: the stream that call named now waits for the launch that wrote it
static int fused_reader_joins(tbx_engine* e)
{
    TbxPipe& p = e->pipe;
    if (!p.active || !p.fused) return TBX_OK;
    const int k = e->out_par;
    if (p.launch_rec[k] && !p.user_waits[k]) {
        EHIP(hipSetDevice(e->device));
        EHIP(hipStreamWaitEvent(e->last_stream, p.launch_ev[k], 0));
        p.user_waits[k] = true;
    }
    p.reader_seen = true;
    return TBX_OK;
}

static bool fused_overlap_on(const tbx_engine* e, const uint8_t* out_dev, int channels)
{
    const int v = e->opt[TBX_OPT_FUSED_OVERLAP];
    if (v == 2 || out_dev != nullptr || !e->ops->render_step_fused(channels)) return false;
    return v == 1 || e->ops->fused_overlap_auto(e->n, !e->gather ? 0 : e->gather_ring ? 2 : 1);
}

static int fused_overlapped(tbx_engine* e, int channels, const ActionSource& src, uint32_t flags, hipStream_t user)
{
    int rc = pipe_enter(e, PIPE_FUSED_OVERLAP);
    if (rc) return rc;
    TbxPipe& p = e->pipe;
    const int cur = e->out_par, wp = cur ^ 1;                  // this launch writes output set wp and frame buffer wp, on lane wp
#ifdef TBX_DIAG
    const int diag = getenv("TBX_OVERLAP_DIAG") ? atoi(getenv("TBX_OVERLAP_DIAG")) : 0;
#else
    const int diag = 0;
#endif
    hipStream_t ls = p.lane[OVL_DIAG(diag, 8) ? 0 : wp];
    const int fb = OVL_DIAG(diag, 32) ? 0 : wp;
    const size_t bytes = (size_t)e->n * e->ops->height() * e->ops->width() * channels;
    if (p.frame_bytes[fb] < bytes) {
        EHIP(hipStreamSynchronize(ls));
        if (p.frame[fb]) hipFree(p.frame[fb]);
        p.frame[fb] = nullptr;
        p.frame_bytes[fb] = 0;
        EHIP(hipMalloc((void**)&p.frame[fb], bytes));
        p.frame_bytes[fb] = bytes;
    }
    // Readers of what call N-2 left in set / frame wp were queued on U before call N-1, which fenced them; this call fences the
    // readers of call N-1's results for call N+1 -- if there can be any: U joined lazily (fused_reader_joins), and a caller
    // that took no address since the last call has queued no reader.
    if (p.user_step_rec[wp]) { EHIP(hipStreamWaitEvent(ls, p.user_step_ev[wp], 0)); p.user_step_rec[wp] = false; }
    if (p.reader_seen || OVL_DIAG(diag, 16)) {
        EHIP(hipEventRecord(p.user_step_ev[cur], user));
        p.user_step_rec[cur] = true;
        p.reader_seen = false;
    }
    uint64_t* const ring_slot = e->gather_ring ? e->packed : nullptr;      // (a K-step ring owns the record pointer)
    tbx_set_out_parity(e, wp);
    if (ring_slot) { e->packed = ring_slot; e->ops->rebind_outputs(e); }
    auto undo = [&]() { tbx_set_out_parity(e, cur); if (ring_slot) { e->packed = ring_slot; e->ops->rebind_outputs(e); } };
    hipError_t ge = tbx_gather_before_step(e, ls);
    if (ge != hipSuccess) { undo(); return hip_fail(e, "tbx_gather_before_step", ge); }
    if (p.prev_overlapped && !OVL_DIAG(diag, 64)) {
        hipLaunchKernelGGL(tbx_ticket_wait_kernel, dim3(1), dim3(64), 0, ls, p.arrive, p.arrive_want, p.release_want, e->err_flag);
        ge = hipGetLastError();
        if (ge != hipSuccess) { undo(); return hip_fail(e, "tbx_ticket_wait_kernel", ge); }
    }
    TbxOverlapLaunch ov{p.arrive, p.launch_ev[wp], e->opt[TBX_OPT_FUSED_OVERLAP_LEAD], 0, diag};
    rc = e->ops->render_step(e, p.frame[fb], channels, src, flags, ls, &ov);
    if (rc) { undo(); return rc; }                             // nothing was launched: TBX_BUF_* keep naming the set that holds results
    p.arrive_want += (unsigned long long)ov.step_blocks;
    p.release_want += 1;
    p.prev_overlapped = true;
    p.launch_rec[wp] = true;
    if (OVL_DIAG(diag, 128)) EHIP(hipEventRecord(p.launch_ev[wp], ls));
    p.user_waits[wp] = false;
    if (OVL_DIAG(diag, 16)) { EHIP(hipStreamWaitEvent(user, p.launch_ev[wp], 0)); p.user_waits[wp] = true; }    // (DIAG: the eager join of the first build)
    e->frame = p.frame[fb];
    e->frame_bytes = p.frame_bytes[fb];
    e->step_carries_order_ev = false;
    e->last_stream = user;
    e->has_last = true;
    return TBX_OK;
}

// ---- rollout chunks (tbx_rollout_synthetic; contract in include/toybox_amd.h, TbxPipe::rollout)
//
// Chunk c (parity q = c & 1) of k frames:
//   step lane   T_c: ONE launch steps every env k frames with the state in registers; it writes the k render records R[q][0..k-1]
//               (record j = the state before frame j), the k step records (straight into a ring of the gather, or an engine-owned
//               [k][N] array) and the state.  It is ordered behind T_{c-1} by the lane, behind the rasterisers of chunk c-2 (they read
//               R[q]), behind the collective that last read the ring, behind the caller's readers of output set q.
//   lanes 0, 1  the rasteriser launches R[q][j] -> F[q][j]: k plain ones, launch j on lane j & 1, or ONE over the chunk's k x N frames
//               on lane 0 (GameOps::rollout_render_span; below).  They wait for T_c and for nothing else -- and T_{c+1} has the whole
//               length of these launches to finish beside them.  No launch ever waits for a step that is
//               running beside a rasteriser (a Breakout step kernel that takes 10 us alone takes 100-250 us there -- what made the
//               device-side ticket of overlapped fused launches wait for most of the launch before it).
//   collective  (K-step ring, K = k) behind T_c alone: it runs beside the chunk's rasterisers.
//   caller      joins lazily (tbx_device_buffer), as with overlapped fused launches.
static bool rollout_chunks_on(const tbx_engine* e, int channels)
{
    const int v = e->opt[TBX_OPT_ROLLOUT_CHUNKS];
    if (v == 2 || !e->ops->rollout_ok(channels)) return false;
    if (e->gather && !e->gather_ring) return false;              // one collective per step: k collectives cannot ride on one launch
    return v == 1 || v >= 3 || e->ops->rollout_auto(e->n, !e->gather ? 0 : e->gather_ring ? 2 : 1);
}

static int chunk_buffers(tbx_engine* e, int q, int k, size_t frame_bytes, bool want_packed, hipStream_t sync_a, hipStream_t sync_b)
{
    TbxPipe& p = e->pipe;
    if (p.chunk_frame_bytes[q] < (size_t)k * frame_bytes) {
        if (sync_a) EHIP(hipStreamSynchronize(sync_a));
        if (sync_b) EHIP(hipStreamSynchronize(sync_b));
        if (p.chunk_frames[q]) hipFree(p.chunk_frames[q]);
        p.chunk_frames[q] = nullptr;
        p.chunk_frame_bytes[q] = 0;
        EHIP(hipMalloc((void**)&p.chunk_frames[q], (size_t)k * frame_bytes));
        p.chunk_frame_bytes[q] = (size_t)k * frame_bytes;
    }
    const size_t pb = sizeof(uint64_t) * (size_t)k * (size_t)e->n;
    if (want_packed && p.chunk_packed_bytes[q] < pb) {
        if (sync_a) EHIP(hipStreamSynchronize(sync_a));
        if (sync_b) EHIP(hipStreamSynchronize(sync_b));
        if (p.chunk_packed[q]) hipFree(p.chunk_packed[q]);
        p.chunk_packed[q] = nullptr;
        p.chunk_packed_bytes[q] = 0;
        EHIP(hipMalloc((void**)&p.chunk_packed[q], pb));
        p.chunk_packed_bytes[q] = pb;
    }
    return TBX_OK;
}

static int rollout_chunked(tbx_engine* e, int channels, const ActionSource& src, uint32_t flags, int k, hipStream_t user)
{
    int rc = pipe_enter(e, PIPE_ROLLOUT_CHUNKS);
    if (rc) return rc;
    TbxPipe& p = e->pipe;
    const int cur = e->out_par, q = cur ^ 1;                   // this chunk: output set q, record buffer q, frame chunk q
    hipStream_t ss = p.step_lane;
    const size_t fb = (size_t)e->n * e->ops->height() * e->ops->width() * channels;
    rc = chunk_buffers(e, q, k, fb, !e->gather_ring, p.lane[0], p.lane[1]);
    if (rc) return rc;
    uint64_t* packed = p.chunk_packed[q];
    size_t stride = (size_t)e->n;
    if (e->gather_ring) {
        rc = tbx_gather_ring_open(e, ss, k, &packed, &stride);
        if (rc) return rc;
    }
    // T_c behind the rasterisers of chunk c - 2, which read the record buffer it rewrites
    for (int l = 0; l < 2; l++)
        if (p.chunk_raster_rec[q][l]) EHIP(hipStreamWaitEvent(ss, p.chunk_raster_ev[q][l], 0));
    // ... and behind the caller's readers of what chunk c - 2 left in output set q (fenced by call c - 1); the same fence holds the
    // rasterisers below back from the frames those readers may still read.  This call fences the readers of chunk c - 1, if the
    // caller took an address since (lazy join: no address, no reader)
    const bool fenced = p.user_step_rec[q];
    if (fenced) { EHIP(hipStreamWaitEvent(ss, p.user_step_ev[q], 0)); p.user_step_rec[q] = false; }
    if (p.reader_seen) {
        EHIP(hipEventRecord(p.user_step_ev[cur], user));
        p.user_step_rec[cur] = true;
        p.reader_seen = false;
    }
    tbx_set_out_parity(e, q);
    e->packed = packed + (size_t)(k - 1) * stride;              // TBX_BUF_PACKED: the record of the chunk's last step
    e->ops->rebind_outputs(e);
    rc = e->ops->rollout_step(e, src, flags, k, q, packed, stride, ss);
    if (rc) { tbx_set_out_parity(e, cur); return rc; }
    EHIP(hipEventRecord(p.chunk_step_ev[q], ss));
    p.chunk_step_rec[q] = true;
    bool lane_used[2] = {false, false};
    auto lane_for = [&](int l) -> int {
        if (!lane_used[l]) {
            EHIP(hipStreamWaitEvent(p.lane[l], p.chunk_step_ev[q], 0));
            if (fenced) EHIP(hipStreamWaitEvent(p.lane[l], p.user_step_ev[q], 0));
            lane_used[l] = true;
        }
        return TBX_OK;
    };
    const int form = e->opt[TBX_OPT_ROLLOUT_CHUNKS];
    const bool span = e->ops->rollout_span_ok() && (form == 4 || (form != 3 && e->ops->rollout_span_auto(e->n, !e->gather ? 0 : 2)));
    if (span) {
        // The chunk's frames in ONE rasteriser launch (k x n frames; at most 65 536 frames per launch), chunk behind chunk on ONE internal
        // stream: rasteriser launches back to back as in a render-only loop, k times as long as a frame's, the next chunk's step launch
        // beside them on the step lane.  Nothing runs beside a rasteriser launch but that step launch, so the rate does not depend on where
        // the frame buffers lie (two rasteriser launches side by side: 0.151-0.164 ms per step at 8 192 envs from process to process,
        // this form 0.1533-0.1536 in every one; profiles/r06_experiments.txt item 6).
        const int l = 0;
        // is the previous chunk's launch still running on that stream?  Then this one starts like a launch of a render-only loop
        bool behind = p.chunk_raster_rec[cur][l] && hipEventQuery(p.chunk_raster_ev[cur][l]) == hipErrorNotReady;
        (void)hipGetLastError();                                // (hipErrorNotReady is an answer, not an error to keep)
        rc = lane_for(l);
        if (rc) return rc;
        const int per = std::max(1, std::min(k, 65536 / std::max(1, e->n)));
        for (int j0 = 0; j0 < k; j0 += per) {
            rc = e->ops->rollout_render_span(e, p.chunk_frames[q] + (size_t)j0 * fb, channels, q, j0, std::min(per, k - j0), behind, p.lane[l]);
            if (rc) return rc;
            behind = true;
        }
    } else {
        for (int j = 0; j < k; j++) {
            const int l = j & 1;
            rc = lane_for(l);
            if (rc) return rc;
            rc = e->ops->rollout_render(e, p.chunk_frames[q] + (size_t)j * fb, channels, q, j, p.lane[l]);
            if (rc) return rc;
        }
    }
    for (int l = 0; l < 2; l++) {
        p.chunk_raster_rec[q][l] = lane_used[l];
        if (lane_used[l]) EHIP(hipEventRecord(p.chunk_raster_ev[q][l], p.lane[l]));
    }
    if (e->gather_ring) {
        rc = tbx_gather_ring_filled(e, p.chunk_step_ev[q]);
        if (rc) return rc;
    }
    p.chunk_user_waits[q] = false;
    p.chunk_cur = q; p.chunk_k = k; p.chunk_channels = channels;
    p.chunk_packed_base = packed; p.chunk_packed_stride = stride;
    e->frame = p.chunk_frames[q] + (size_t)(k - 1) * fb;        // TBX_BUF_FRAME: the chunk's last frame
    e->frame_bytes = fb;
    e->step_carries_order_ev = false;
    e->last_stream = user;
    e->has_last = true;
    return TBX_OK;
}

// the caller asks where a result of the last chunk lies: the stream that call named now waits for the chunk (its step launch and,
// for the frames, its rasterisers)
static int rollout_reader_joins(tbx_engine* e, bool frames)
{
    TbxPipe& p = e->pipe;
    if (!p.active || !p.rollout) return TBX_OK;
    const int q = p.chunk_cur;
    EHIP(hipSetDevice(e->device));
    if (p.chunk_step_rec[q] && !p.chunk_user_waits[q]) EHIP(hipStreamWaitEvent(e->last_stream, p.chunk_step_ev[q], 0));
    if (frames)
        for (int l = 0; l < 2; l++)
            if (p.chunk_raster_rec[q][l]) EHIP(hipStreamWaitEvent(e->last_stream, p.chunk_raster_ev[q][l], 0));
    p.chunk_user_waits[q] = true;
    p.reader_seen = true;
    return TBX_OK;
}

int tbx_step_synthetic(tbx_engine* e, uint64_t action_seed, uint64_t t, uint64_t env_offset, uint32_t flags, void* stream)
{
    CHECK_ENGINE(e);
    EHIP(hipSetDevice(e->device));
    ActionSource src{};
    src.actions = nullptr;
    src.seed = action_seed;
    src.t = t;
    src.env_offset = env_offset;
    src.single_env = -1;
    if (const int mode = pipe_mode(e)) return pipe_step(e, src, flags, (hipStream_t)stream, mode);
    EHIP(tbx_use_stream(e, (hipStream_t)stream));
    EHIP(tbx_gather_before_step(e, (hipStream_t)stream));
    return e->ops->step(e, src, flags, (hipStream_t)stream);
}

// ---- host delivery (toybox_amd.h: step_async / step_wait).  tbx_step is begin + end without a frame.

int tbx_host_alloc(void** out_ptr, size_t bytes)
{
    if (!out_ptr) { tbx_set_create_error("tbx_host_alloc: out_ptr is NULL"); return TBX_E_INVALID; }
    *out_ptr = nullptr;
    hipError_t r = hipHostMalloc(out_ptr, bytes ? bytes : 1, hipHostMallocDefault);
    if (r != hipSuccess) { tbx_set_create_error(std::string("hipHostMalloc: ") + hipGetErrorString(r)); return r == hipErrorOutOfMemory ? TBX_E_NOMEM : TBX_E_NO_DEVICE; }
    return TBX_OK;
}

int tbx_host_free(void* ptr)
{
    if (!ptr) return TBX_OK;
    hipError_t r = hipHostFree(ptr);
    if (r != hipSuccess) { tbx_set_create_error(std::string("hipHostFree: ") + hipGetErrorString(r)); return TBX_E_INVALID; }
    return TBX_OK;
}

// rows [first, last) of the host-side frame stack (see toybox_amd.h)
static void host_stack_rows(uint8_t* dst, const uint8_t* src, const uint8_t* plane, const uint8_t* done, int reset, int first, int last,
                            int px, int stack, int fill)
{
    for (int i = first; i < last; i++) {
        uint8_t* d = dst + (size_t)i * px * stack;
        const uint8_t* s = src + (size_t)i * px * stack;
        const uint8_t* p = plane + (size_t)i * px;
        const bool fresh = reset || (done && done[i]);
        if (stack == 4) {                                    // one dword per pixel: shift the older three down, the new byte on top
            uint32_t* d4 = reinterpret_cast<uint32_t*>(d);
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
            if (d != s && (px & 15) == 0 && ((((uintptr_t)d | (uintptr_t)s | (uintptr_t)p)) & 15u) == 0) {
                // 16 pixels per turn with streaming stores: the destination (another array of the pool) is written without being
                // read first -- 8 bytes of host memory traffic per pixel instead of 12 (the roll is bound by exactly that)
                const __m128i zero = _mm_setzero_si128();
                for (int k = 0; k < px; k += 16) {
                    const __m128i pb = _mm_load_si128(reinterpret_cast<const __m128i*>(p + k));
                    const __m128i lo16 = _mm_unpacklo_epi8(zero, pb), hi16 = _mm_unpackhi_epi8(zero, pb);   // byte -> high half of a word
                    const __m128i top[4] = {_mm_unpacklo_epi16(zero, lo16), _mm_unpackhi_epi16(zero, lo16),   // ... -> top byte of a dword
                                            _mm_unpacklo_epi16(zero, hi16), _mm_unpackhi_epi16(zero, hi16)};
                    for (int q = 0; q < 4; q++) {
                        __m128i v;
                        if (!fresh) v = _mm_or_si128(_mm_srli_epi32(_mm_load_si128(reinterpret_cast<const __m128i*>(s4 + k + 4 * q)), 8), top[q]);
                        else if (fill) { const __m128i b = _mm_srli_epi32(top[q], 24); v = _mm_or_si128(_mm_or_si128(b, _mm_slli_epi32(b, 8)), _mm_or_si128(_mm_slli_epi32(b, 16), top[q])); }
                        else v = top[q];
                        _mm_stream_si128(reinterpret_cast<__m128i*>(d4 + k + 4 * q), v);
                    }
                }
                continue;
            }
            if (!fresh) for (int k = 0; k < px; k++) d4[k] = (s4[k] >> 8) | ((uint32_t)p[k] << 24);
            else if (fill) for (int k = 0; k < px; k++) d4[k] = (uint32_t)p[k] * 0x01010101u;
            else for (int k = 0; k < px; k++) d4[k] = (uint32_t)p[k] << 24;
            continue;
        }
        for (int k = 0; k < px; k++) {
            uint8_t* dk = d + (size_t)k * stack;
            const uint8_t* sk = s + (size_t)k * stack;
            for (int c = 0; c + 1 < stack; c++) dk[c] = fresh ? (fill ? p[k] : (uint8_t)0) : sk[c + 1];
            dk[stack - 1] = p[k];
        }
    }
    _mm_sfence();                                            // the streaming stores above are globally visible before the thread is joined
}

int tbx_host_stack_push(uint8_t* dst, const uint8_t* src, const uint8_t* plane, const uint8_t* done, int reset, int n, int px, int stack,
                        int fill, int threads)
{
    if (!dst || !src || !plane || n < 0 || px < 1 || stack < 1 || stack > 16 || fill < 0 || fill > 1) {
        tbx_set_create_error("tbx_host_stack_push: bad argument");
        return TBX_E_INVALID;
    }
    if (stack == 4 && ((((uintptr_t)dst | (uintptr_t)src) & 3u) != 0)) { tbx_set_create_error("tbx_host_stack_push: stacks of depth 4 must be 4-byte aligned"); return TBX_E_INVALID; }
    int T = threads;
    if (T <= 0) {
        cpu_set_t set;
        T = sched_getaffinity(0, sizeof set, &set) == 0 ? CPU_COUNT(&set) : 1;
        if (T > 16) T = 16;
    }
    const long work = (long)n * px;
    if ((long)T > work >> 19) T = (int)(work >> 19);         // half a million pixels per thread at least: starting a thread costs ~20 us
                                                             // (64 envs: 0.54 ms with sixteen threads, 0.07 with one)
    if (T > n) T = n > 0 ? n : 1;
    if (T <= 1) { host_stack_rows(dst, src, plane, done, reset, 0, n, px, stack, fill); return TBX_OK; }
    std::vector<std::thread> pool;
    pool.reserve((size_t)T);
    for (int t = 0; t < T; t++) {
        const int first = (int)((long)n * t / T), last = (int)((long)n * (t + 1) / T);
        pool.emplace_back(host_stack_rows, dst, src, plane, done, reset, first, last, px, stack, fill);
    }
    for (auto& th : pool) th.join();
    return TBX_OK;
}

// the waiting half of tbx_step_end: the queued step has finished, its outputs are in the caller's buffers
static int step_deliver(tbx_engine* e)
{
    EHIP(hipSetDevice(e->device));
    EHIP(hipStreamSynchronize(e->stream));
    e->host_pending = false;
    e->pending_kind = 0;
    const size_t N = (size_t)e->n;
    const int32_t* host_out = e->io_host + N;
    const tbx_step_host_out_t& o = e->host_out;
    if (o.reward) memcpy(o.reward, host_out, N * sizeof(int32_t));
    if (o.lives) memcpy(o.lives, host_out + N, N * sizeof(int32_t));
    if (o.score) memcpy(o.score, host_out + 2 * N, N * sizeof(int32_t));
    if (o.done) memcpy(o.done, host_out + 3 * N + 1, N);
    if (host_out[3 * N] & 2) return e->fail(TBX_E_NEEDS_RESET, "an env was stepped after its game ended inside EpisodicLifeEnv's no-op step (bench.Monitor raises here)");
    if (host_out[3 * N]) return e->fail(TBX_E_ACTION, "an illegal ALE action id was passed (treated as NOOP)");
    return TBX_OK;
}

// Another entry point was called between "_begin" and "_end" (every one that queues work comes through tbx_use_stream or
// pipe_enter): the step is ended here -- the stream drained, the outputs delivered to the caller's buffers -- and what its "_end"
// call would have returned is kept for that call (the header promised this, and a later "_end" delivered stale data).
}  // extern "C"

hipError_t tbx_finish_pending(tbx_engine* e)
{
    const int kind = e->pending_kind;
    if (!kind) return hipSuccess;
    const int rc = kind == 1 ? step_deliver(e) : tbx_agent_deliver(e);
    e->pending_kind = 0;
    e->ended_early_kind = kind;
    e->ended_early_rc = rc;
    e->ended_early_msg = rc ? e->err : std::string();
    return rc == TBX_E_NO_DEVICE ? hipErrorUnknown : hipSuccess;
}

extern "C" {

int tbx_step_begin(tbx_engine* e, const int32_t* actions_host, uint32_t flags, const tbx_step_host_out_t* out)
{
    CHECK_ENGINE(e);
    if (!actions_host || !out) return e->fail(TBX_E_INVALID, "actions / output descriptor is NULL");
    if (e->pending_kind) return e->fail(TBX_E_INVALID, "tbx_step_begin: the previous step has not been ended (tbx_step_end / tbx_agent_step_end)");
    e->ended_early_kind = 0;                   // (an "_end" that was never called for a step another call ended)
    if (out->frame && out->channels != 1 && out->channels != 3 && out->channels != 4) return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    EHIP(tbx_gather_before_step(e, e->stream));
    const size_t N = (size_t)e->n;
    const size_t out_bytes = (3 * N + 1) * sizeof(int32_t) + N;
    if (!e->io_dev) {
        EHIP(hipMalloc((void**)&e->io_dev, out_bytes));
        EHIP(hipHostMalloc((void**)&e->io_host, N * sizeof(int32_t) + out_bytes, hipHostMallocDefault));
    }
    int32_t* host_actions = e->io_host;
    int32_t* host_out = e->io_host + N;
    memcpy(host_actions, actions_host, N * sizeof(int32_t));
    EHIP(hipMemcpyAsync(e->actions, host_actions, N * sizeof(int32_t), hipMemcpyHostToDevice, e->stream));
    ActionSource src{};
    src.actions = e->actions;
    src.single_env = -1;
    int rc = e->ops->step(e, src, flags, e->stream);
    if (rc) return rc;
    hipLaunchKernelGGL(gather_outputs_kernel, dim3((e->n + 255) / 256), dim3(256), 0, e->stream, e->reward, e->lives_out, e->score_out,
                       e->done, e->err_flag, e->io_dev, e->n);
    EHIP(hipGetLastError());
    EHIP(hipMemcpyAsync(host_out, e->io_dev, out_bytes, hipMemcpyDeviceToHost, e->stream));
    if (out->frame) {
        const size_t bytes = N * e->ops->height() * e->ops->width() * out->channels;
        rc = ensure_frame(e, bytes);
        if (rc) return rc;
        rc = e->ops->render(e, e->frame_own, out->channels, 0, e->n, e->stream);
        if (rc) return rc;
        e->frame = e->frame_own;
        e->frame_bytes = e->frame_own_bytes;
        EHIP(hipMemcpyAsync(out->frame, e->frame_own, bytes, hipMemcpyDeviceToHost, e->stream));
    }
    e->host_out = *out;
    e->host_pending = true;
    e->pending_kind = 1;
    return TBX_OK;
}

int tbx_step_end(tbx_engine* e)
{
    CHECK_ENGINE(e);
    if (!e->host_pending) {
        if (e->ended_early_kind == 1) {        // another call on the handle ended the step: its outputs have been delivered
            e->ended_early_kind = 0;
            return e->ended_early_rc ? e->fail(e->ended_early_rc, e->ended_early_msg) : TBX_OK;
        }
        return e->fail(TBX_E_INVALID, "tbx_step_end without tbx_step_begin");
    }
    return step_deliver(e);
}

int tbx_step(tbx_engine* e, const int32_t* actions_host, uint32_t flags, int32_t* reward, uint8_t* done,
             int32_t* lives, int32_t* score)
{
    CHECK_ENGINE(e);
    if (e->pending_kind) return e->fail(TBX_E_INVALID, "tbx_step: a step is between its begin and end calls (tbx_step_end / tbx_agent_step_end)");
    tbx_step_host_out_t out{};
    out.reward = reward; out.done = done; out.lives = lives; out.score = score;
    int rc = tbx_step_begin(e, actions_host, flags, &out);
    if (rc) return rc;
    return tbx_step_end(e);
}

// ---- resident single-env step (TbxServeCtl, tbx_common.hpp)

// the pinned, device-mapped frame buffer of the single-env path (tbx_step1_frame): the resident kernel rasterises into it
static int ensure_serve_frame(tbx_engine* e)
{
    if (e->serve_frame) return TBX_OK;
    const size_t bytes = (size_t)e->ops->height() * e->ops->width() * 4;
    EHIP(hipHostMalloc((void**)&e->serve_frame, bytes, hipHostMallocMapped));
    memset(e->serve_frame, 0, bytes);
    EHIP(hipHostGetDevicePointer((void**)&e->serve_frame_dev, e->serve_frame, 0));
    return TBX_OK;
}

static int serve_start(tbx_engine* e)
{
    if (!e->serve_ctl) {
        EHIP(hipHostMalloc((void**)&e->serve_ctl, sizeof(TbxServeCtl), hipHostMallocMapped | hipHostMallocCoherent));
        memset(e->serve_ctl, 0, sizeof(TbxServeCtl));
        EHIP(hipHostGetDevicePointer((void**)&e->serve_ctl_dev, e->serve_ctl, 0));
        EHIP(hipStreamCreateWithFlags(&e->serve_stream, hipStreamNonBlocking));
    }
    if (!e->serve_frame) {
        int rc = ensure_serve_frame(e);
        if (rc) return rc;
    }
    TbxServeCtl* c = e->serve_ctl;
    c->exited = 0;
    c->frame_dev = (uint64_t)(uintptr_t)e->serve_frame_dev;
    c->ack_seq = e->serve_seq;
    c->req = (uint64_t)e->serve_seq;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    // ordered after everything queued through the handle so far
    EHIP(tbx_use_stream(e, e->serve_stream));
    int rc = e->ops->serve(e, e->serve_ctl_dev, e->serve_stream);
    if (rc) return rc;
    e->serve_running = true;
    return TBX_OK;
}

}  // extern "C"

hipError_t tbx_serve_stop(tbx_engine* e)
{
    if (!e->serve_running) return hipSuccess;
    __atomic_store_n(&e->serve_ctl->req, (uint64_t)e->serve_seq | TBX_SERVE_STOP, __ATOMIC_RELEASE);
    hipError_t r = hipStreamSynchronize(e->serve_stream);     // the wave leaves at its next poll
    e->serve_running = false;
    return r;
}

extern "C" {

// one frame of one env, optionally with its picture (channels 0 / 1 / 3 / 4) in e->serve_frame
static int step1_impl(tbx_engine* e, int env, int32_t ale_action, uint32_t flags, int channels, int32_t out[4])
{
    CHECK_ENGINE(e);
    if (env < 0 || env >= e->n) return e->fail(TBX_E_INVALID, "env index out of range");
    if (channels != 0 && channels != 1 && channels != 3 && channels != 4) return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
    EHIP(hipSetDevice(e->device));
    bool painted = false;
    if (e->n == 1 && e->opt[TBX_OPT_RESIDENT_STEP] && !e->gather) {
        // the resident kernel: post the request, spin on the acknowledgement (no launch, no copy, no synchronisation)
        TbxServeCtl* c = e->serve_ctl;
        if (!e->serve_running || __atomic_load_n(&c->exited, __ATOMIC_ACQUIRE)) {
            if (e->serve_running) { EHIP(hipStreamSynchronize(e->serve_stream)); e->serve_running = false; }
            int rc = serve_start(e);
            if (rc == TBX_E_UNSUPPORTED) goto slow;
            if (rc) return rc;
            c = e->serve_ctl;
        }
        {
            const uint32_t want = (channels && e->ops->serve_paints()) ? (channels == 1 ? 1u : channels == 3 ? 2u : 3u) : 0u;
            const uint32_t seq = ++e->serve_seq;
            const uint64_t word = tbx_serve_word(seq, ale_action, (flags & 0x0Fu) | (want << TBX_SERVE_FRAME_SHIFT));
            __atomic_store_n(&c->req, word, __ATOMIC_RELEASE);
            unsigned long spins = 0;
            while (__atomic_load_n(&c->ack_seq, __ATOMIC_ACQUIRE) != seq) {
                if ((++spins & 0xFFFul) == 0) {
                    if (__atomic_load_n(&c->exited, __ATOMIC_ACQUIRE) && __atomic_load_n(&c->ack_seq, __ATOMIC_ACQUIRE) != seq) {
                        // the wave went idle and left just as the request arrived: start another, it serves the pending request
                        EHIP(hipStreamSynchronize(e->serve_stream));
                        e->serve_running = false;
                        e->serve_seq = seq - 1;
                        int rc = serve_start(e);
                        if (rc) return rc;
                        e->serve_seq = seq;
                        __atomic_store_n(&c->req, word, __ATOMIC_RELEASE);
                    }
                    if (spins > (1ul << 33)) return e->fail(TBX_E_NO_DEVICE, "the resident step kernel does not answer");
                }
                __builtin_ia32_pause();
            }
            const uint32_t de = c->done_err;
            if (out) { out[0] = c->reward; out[1] = (int32_t)(de & 1u); out[2] = c->lives; out[3] = c->score; }
            painted = want != 0 && !(de & 4u);
            if (channels && !painted) {                      // this game's resident kernel does not paint: the launched rasteriser
                int rc = ensure_serve_frame(e);
                if (rc) return rc;
                rc = tbx_render_env(e, env, e->serve_frame, channels);
                if (rc) return rc;
            }
            if (de & 2u) return e->fail(TBX_E_ACTION, "an illegal ALE action id was passed (treated as NOOP)");
            return TBX_OK;
        }
    }
slow:
    {
        // any env of a batch engine: the single-env launch of tbx_apply_input, then its outputs
        const uint32_t b = tbx_ale_buttons(ale_action);
        const int action_rc = b == 0xFFu ? TBX_E_ACTION : TBX_OK;
        int rc = tbx_apply_input(e, env, b == 0xFFu ? 0u : b);
        if (rc) return rc;
        bool reset = false;
        int32_t o[4] = {0, 0, 0, 0};
        uint8_t dn = 0;
        EHIP(hipMemcpy(&o[0], e->reward + env, 4, hipMemcpyDeviceToHost));
        EHIP(hipMemcpy(&dn, e->done + env, 1, hipMemcpyDeviceToHost));
        EHIP(hipMemcpy(&o[2], e->lives_out + env, 4, hipMemcpyDeviceToHost));
        EHIP(hipMemcpy(&o[3], e->score_out + env, 4, hipMemcpyDeviceToHost));
        o[1] = dn;
        if ((flags & TBX_STEP_AUTO_RESET) && o[2] <= 0) {
            // (rare on this path) a finished game starts over, as in tbx_step
            std::vector<uint8_t> mask((size_t)e->n, 0);
            mask[(size_t)env] = 1;
            rc = tbx_new_game(e, mask.data());
            if (rc) return rc;
            o[1] = 1;
            reset = true;
        }
        (void)reset;
        if (out) memcpy(out, o, sizeof o);
        if (channels) {
            rc = ensure_serve_frame(e);
            if (rc) return rc;
            rc = tbx_render_env(e, env, e->serve_frame, channels);
            if (rc) return rc;
        }
        return action_rc ? e->fail(TBX_E_ACTION, "an illegal ALE action id was passed (treated as NOOP)") : TBX_OK;
    }
}

int tbx_step1(tbx_engine* e, int env, int32_t ale_action, uint32_t flags, int32_t out[4])
{
    return step1_impl(e, env, ale_action, flags, 0, out);
}

int tbx_step1_frame(tbx_engine* e, int env, int32_t ale_action, uint32_t flags, int channels, int32_t out[4], const uint8_t** frame_host)
{
    if (e && channels == 0) return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
    int rc = step1_impl(e, env, ale_action, flags, channels, out);
    if (frame_host && e) *frame_host = (rc == TBX_OK || rc == TBX_E_ACTION) ? e->serve_frame : nullptr;
    return rc;
}

int tbx_apply_input(tbx_engine* e, int env, uint32_t buttons)
{
    CHECK_ENGINE(e);
    if (env < 0 || env >= e->n) return e->fail(TBX_E_INVALID, "env index out of range");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    EHIP(tbx_gather_before_step(e, e->stream));
    ActionSource src{};
    src.single_env = env;
    src.single_buttons = buttons & 0x3Fu;
    int rc = e->ops->step(e, src, 0, e->stream);
    if (rc) return rc;
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_get_scalars(tbx_engine* e, int32_t* score, int32_t* lives, int32_t* level, uint8_t* game_over)
{
    CHECK_ENGINE(e);
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    const size_t N = (size_t)e->n;
    int32_t* tmp = e->scal;
    int rc = e->ops->scalars(e, tmp, tmp + N, tmp + 2 * N, e->stream);
    if (rc) return rc;
    if (!e->scal_host) EHIP(hipHostMalloc((void**)&e->scal_host, 3 * N * sizeof(int32_t), hipHostMallocDefault));
    const int32_t* host = e->scal_host;
    EHIP(hipMemcpyAsync(e->scal_host, tmp, 3 * N * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    EHIP(hipStreamSynchronize(e->stream));
    for (size_t i = 0; i < N; i++) {
        if (score) score[i] = host[i];
        if (lives) lives[i] = host[N + i];
        if (level) level[i] = host[2 * N + i];
        if (game_over) game_over[i] = host[N + i] <= 0;
    }
    return TBX_OK;
}

int tbx_render_device(tbx_engine* e, uint8_t* out_dev, int channels, void* stream)
{
    CHECK_ENGINE(e);
    if (channels != 1 && channels != 3 && channels != 4) return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
    EHIP(hipSetDevice(e->device));
    if (const int mode = pipe_mode(e)) return pipe_render(e, out_dev, channels, (hipStream_t)stream, mode);
    EHIP(tbx_use_stream(e, (hipStream_t)stream));
    if (!out_dev) {
        const size_t bytes = (size_t)e->n * e->ops->height() * e->ops->width() * channels;
        int rc = ensure_frame(e, bytes);
        if (rc) return rc;
        out_dev = e->frame;
    }
    if (((uintptr_t)out_dev & 15u) != 0) return e->fail(TBX_E_INVALID, "frame buffer must be 16-byte aligned");
    return e->ops->render(e, out_dev, channels, 0, e->n, (hipStream_t)stream);
}

int tbx_render_step_synthetic(tbx_engine* e, uint8_t* out_dev, int channels, uint64_t action_seed, uint64_t t, uint64_t env_offset,
                              uint32_t flags, void* stream)
{
    CHECK_ENGINE(e);
    if (channels != 1 && channels != 3 && channels != 4) return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
    EHIP(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    ActionSource src{};
    src.actions = nullptr;
    src.seed = action_seed;
    src.t = t;
    src.env_offset = env_offset;
    src.single_env = -1;
    if (fused_overlap_on(e, out_dev, channels)) return fused_overlapped(e, channels, src, flags, s);
    EHIP(tbx_use_stream(e, s));                                 // (everything on the caller's stream: leaves a pipelined mode first)
    if (!out_dev) {
        const size_t bytes = (size_t)e->n * e->ops->height() * e->ops->width() * channels;
        int rc = ensure_frame(e, bytes);
        if (rc) return rc;
        out_dev = e->frame;                                     // (ensure_frame: TBX_BUF_FRAME is the engine's own buffer again)
    }
    if (((uintptr_t)out_dev & 15u) != 0) return e->fail(TBX_E_INVALID, "frame buffer must be 16-byte aligned");
    if (e->ops->render_step_fused(channels)) {
        EHIP(tbx_gather_before_step(e, s));
        return e->ops->render_step(e, out_dev, channels, src, flags, s);
    }
    // engines whose rasteriser reads live state (or gray frames): the same two things as two launches in stream order
    int rc = e->ops->render(e, out_dev, channels, 0, e->n, s);
    if (rc) return rc;
    EHIP(tbx_gather_before_step(e, s));
    return e->ops->step(e, src, flags, s);
}

int tbx_rollout_synthetic(tbx_engine* e, int channels, uint64_t action_seed, uint64_t t0, int k, uint64_t env_offset, uint32_t flags, void* stream)
{
    CHECK_ENGINE(e);
    if (channels != 1 && channels != 3 && channels != 4) return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
    if (k < 1 || k > 64) return e->fail(TBX_E_INVALID, "tbx_rollout_synthetic: k must be in 1 .. 64");
    EHIP(hipSetDevice(e->device));
    hipStream_t s = (hipStream_t)stream;
    if (e->gather_ring && e->gather_ring_every != k)
        return e->fail(TBX_E_INVALID, "tbx_rollout_synthetic: the chunk must be as long as the gather's record ring (TBX_OPT_GATHER_EVERY)");
    if (e->gather_ring && tbx_gather_fill(e) != 0)
        return e->fail(TBX_E_INVALID, "tbx_rollout_synthetic: the record ring is partly filled (finish it with single steps + tbx_gather)");
    ActionSource src{};
    src.actions = nullptr;
    src.seed = action_seed;
    src.t = t0;
    src.env_offset = env_offset;
    src.single_env = -1;
    if (rollout_chunks_on(e, channels)) return rollout_chunked(e, channels, src, flags, k, s);
    // everywhere else: the k single calls in stream order, frames into chunk buffer 0, the step records into the ring (K-step ring:
    // tbx_gather after every step, the k-th one sends it) or copied out of TBX_BUF_PACKED after every step
    EHIP(tbx_use_stream(e, s));
    TbxPipe& p = e->pipe;
    const size_t fb = (size_t)e->n * e->ops->height() * e->ops->width() * channels;
    int rc = chunk_buffers(e, 0, k, fb, !e->gather_ring, s, nullptr);
    if (rc) return rc;
    uint64_t* ring_base = nullptr;
    for (int j = 0; j < k; j++) {
        rc = tbx_render_step_synthetic(e, p.chunk_frames[0] + (size_t)j * fb, channels, action_seed, t0 + (uint64_t)j, env_offset, flags, s);
        if (rc) return rc;
        if (e->gather_ring) {
            if (j == 0) ring_base = e->packed;                  // slot 0 of the ring this chunk fills
            rc = tbx_gather(e, nullptr, s);
            if (rc) return rc;
        } else if (e->gather) {
            rc = tbx_gather(e, nullptr, s);                     // one collective per step
            if (rc) return rc;
            EHIP(hipMemcpyAsync(p.chunk_packed[0] + (size_t)j * e->n, e->packed, sizeof(uint64_t) * (size_t)e->n, hipMemcpyDeviceToDevice, s));
        } else
            EHIP(hipMemcpyAsync(p.chunk_packed[0] + (size_t)j * e->n, e->packed, sizeof(uint64_t) * (size_t)e->n, hipMemcpyDeviceToDevice, s));
    }
    p.chunk_cur = 0; p.chunk_k = k; p.chunk_channels = channels;
    p.chunk_packed_base = e->gather_ring ? ring_base : p.chunk_packed[0];
    p.chunk_packed_stride = e->gather_ring ? (size_t)e->gather_ring_width : (size_t)e->n;
    return TBX_OK;
}

int tbx_render(tbx_engine* e, uint8_t* out_host, int channels)
{
    CHECK_ENGINE(e);
    if (!out_host) return e->fail(TBX_E_INVALID, "output pointer is NULL");
    int rc = tbx_render_device(e, nullptr, channels, e->stream);
    if (rc) return rc;
    const size_t bytes = (size_t)e->n * e->ops->height() * e->ops->width() * channels;
    EHIP(hipMemcpyAsync(out_host, e->frame, bytes, hipMemcpyDeviceToHost, e->stream));
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_render_env(tbx_engine* e, int env, uint8_t* out_host, int channels)
{
    CHECK_ENGINE(e);
    if (env < 0 || env >= e->n || !out_host) return e->fail(TBX_E_INVALID, "env index out of range");
    if (channels != 1 && channels != 3 && channels != 4) return e->fail(TBX_E_INVALID, "channels must be 1, 3 or 4");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    const size_t bytes = (size_t)e->ops->height() * e->ops->width() * channels;
    if (!e->one_frame) EHIP(hipMalloc((void**)&e->one_frame, (size_t)e->ops->height() * e->ops->width() * 4));
    int rc = e->ops->render(e, e->one_frame, channels, env, 1, e->stream);
    if (rc) return rc;
    EHIP(hipMemcpyAsync(out_host, e->one_frame, bytes, hipMemcpyDeviceToHost, e->stream));
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

static int ensure_staging(tbx_engine* e, size_t bytes)
{
    if (e->staging_bytes >= bytes) return TBX_OK;
    EHIP(hipStreamSynchronize(e->stream));
    if (e->staging) hipFree(e->staging);
    e->staging = nullptr;
    e->staging_bytes = 0;
    EHIP(hipMalloc(&e->staging, bytes));
    e->staging_bytes = bytes;
    return TBX_OK;
}

int tbx_get_states(tbx_engine* e, int first_env, int count, void* pods, size_t record_size)
{
    CHECK_ENGINE(e);
    if (first_env < 0 || count < 1 || first_env + count > e->n || !pods) return e->fail(TBX_E_INVALID, "env range out of bounds");
    if (record_size != e->ops->state_size()) return e->fail(TBX_E_INVALID, "state record size mismatch");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    int rc = ensure_staging(e, record_size * (size_t)count);
    if (rc) return rc;
    rc = e->ops->pack_state(e, first_env, count, e->stream);
    if (rc) return rc;
    EHIP(hipMemcpyAsync(pods, e->staging, record_size * (size_t)count, hipMemcpyDeviceToHost, e->stream));
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_set_states(tbx_engine* e, int first_env, int count, const void* pods, size_t record_size)
{
    CHECK_ENGINE(e);
    if (first_env < 0 || count < 1 || first_env + count > e->n || !pods) return e->fail(TBX_E_INVALID, "env range out of bounds");
    if (record_size != e->ops->state_size()) return e->fail(TBX_E_INVALID, "state record size mismatch");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    int rc = ensure_staging(e, record_size * (size_t)count);
    if (rc) return rc;
    rc = e->ops->unpack_state(e, first_env, count, pods, e->stream);
    if (rc) return rc;
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_get_state(tbx_engine* e, int env, void* pod, size_t size)
{
    CHECK_ENGINE(e);
    if (env < 0 || env >= e->n || !pod) return e->fail(TBX_E_INVALID, "env index out of range");
    return tbx_get_states(e, env, 1, pod, size);
}

int tbx_set_state(tbx_engine* e, int env, const void* pod, size_t size)
{
    CHECK_ENGINE(e);
    if (env < 0 || env >= e->n || !pod) return e->fail(TBX_E_INVALID, "env index out of range");
    return tbx_set_states(e, env, 1, pod, size);
}

int tbx_get_config(tbx_engine* e, void* pod, size_t size)
{
    CHECK_ENGINE(e);
    if (!pod || size != e->ops->config_size()) return e->fail(TBX_E_INVALID, "config record size mismatch");
    int rc = e->ops->get_config(e, pod);
    if (rc) return rc;
    uint64_t r[2];
    rc = tbx_get_sim_rng(e, 0, r);
    if (rc) return rc;
    memcpy(pod, r, sizeof r);
    return TBX_OK;
}

int tbx_set_config(tbx_engine* e, const void* pod, size_t size)
{
    CHECK_ENGINE(e);
    if (!pod || size != e->ops->config_size()) return e->fail(TBX_E_INVALID, "config record size mismatch");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    EHIP(hipStreamSynchronize(e->stream));
    int rc = e->ops->set_config(e, pod);
    if (rc) return rc;
    // `rand` as tbx_get_config reported it (env 0's words) means "not edited": the per-env simulator RNGs stay as they are,
    // so a config edit does not collapse a seeded batch onto one stream.  Any other value is written to every env.
    uint64_t r[2], cur[2];
    memcpy(r, pod, sizeof r);
    rc = tbx_get_sim_rng(e, 0, cur);
    if (rc) return rc;
    if (r[0] == cur[0] && r[1] == cur[1]) return TBX_OK;
    return tbx_set_sim_rng(e, -1, r);
}

// ---- batched interventions (include/toybox_amd.h): field writes and per-env features over the state in HBM

int tbx_reduce_width(int game, int query)
{
    switch (query) {
    case TBX_QUERY_BRK_BRICKS_REMAINING: case TBX_QUERY_BRK_NUM_BRICKS: case TBX_QUERY_BRK_IS_CHANNEL: case TBX_QUERY_BRK_CHANNEL_COUNT:
    case TBX_QUERY_BRK_FIND_CHANNEL: return game == TBX_GAME_BREAKOUT ? 1 : TBX_E_INVALID;
    case TBX_QUERY_BRK_COLUMN: case TBX_QUERY_BRK_ROW: return game == TBX_GAME_BREAKOUT ? 32 : TBX_E_INVALID;
    case TBX_QUERY_BRK_PADDLE: return game == TBX_GAME_BREAKOUT ? 4 : TBX_E_INVALID;
    case TBX_QUERY_BRK_BALLS: return game == TBX_GAME_BREAKOUT ? 1 + 4 * TBX_BRK_MAX_BALLS : TBX_E_INVALID;
    case TBX_QUERY_AMI_MODE: return game == TBX_GAME_AMIDAR ? 2 : TBX_E_INVALID;
    case TBX_QUERY_AMI_ANY_CAUGHT: case TBX_QUERY_AMI_TILE: case TBX_QUERY_AMI_COUNT_TILES: case TBX_QUERY_AMI_PLAYER_ON_PAINTED:
    case TBX_QUERY_AMI_PLAYER_NEAR_UNPAINTED: return game == TBX_GAME_AMIDAR ? 1 : TBX_E_INVALID;
    case TBX_QUERY_AMI_ADJACENT: return game == TBX_GAME_AMIDAR ? 4 : TBX_E_INVALID;
    case TBX_QUERY_AMI_ENEMY_DISTANCES: case TBX_QUERY_AMI_PLAYER_ENEMY_DISTANCES: return game == TBX_GAME_AMIDAR ? TBX_AMI_MAX_ENEMIES : TBX_E_INVALID;
    case TBX_QUERY_AMI_PLAYER_TILE: return game == TBX_GAME_AMIDAR ? 3 : TBX_E_INVALID;
    case TBX_QUERY_SI_SHIP: return game == TBX_GAME_SPACE_INVADERS ? 8 : TBX_E_INVALID;
    case TBX_QUERY_BRK_FIND_BRICK: return game == TBX_GAME_BREAKOUT ? 1 : TBX_E_INVALID;
    case TBX_QUERY_AMI_TILES_MASK: return game == TBX_GAME_AMIDAR ? 32 : TBX_E_INVALID;
    case TBX_QUERY_AMI_RANDOM_TILE: return game == TBX_GAME_AMIDAR ? 4 : TBX_E_INVALID;
    case TBX_QUERY_AMI_RANDOM_DIR: return game == TBX_GAME_AMIDAR ? 2 : TBX_E_INVALID;
    default: return TBX_E_INVALID;
    }
}

static int edit_args(tbx_engine* e, const double* args, int n_args, int per_env, bool args_on_host, hipStream_t s, TbxEditArgs& a)
{
    if (n_args < 0 || n_args > TBX_EDIT_MAX_ARGS || (n_args > 0 && !args)) return e->fail(TBX_E_INVALID, "bad intervention arguments");
    memset(&a, 0, sizeof a);
    a.n = n_args;
    a.per_env = nullptr;
    if (!per_env || n_args == 0) {
        for (int i = 0; i < n_args; i++) a.v[i] = args[i];                  // (a host pointer in every form)
        return TBX_OK;
    }
    if (!args_on_host) { a.per_env = args; return TBX_OK; }
    const size_t bytes = sizeof(double) * (size_t)e->n * (size_t)n_args;
    if (e->edit_args_bytes < bytes) {
        EHIP(hipStreamSynchronize(e->stream));
        hipFree(e->edit_args);
        e->edit_args = nullptr; e->edit_args_bytes = 0;
        EHIP(hipMalloc((void**)&e->edit_args, bytes));
        e->edit_args_bytes = bytes;
    }
    EHIP(hipMemcpyAsync(e->edit_args, args, bytes, hipMemcpyHostToDevice, s));
    a.per_env = e->edit_args;
    return TBX_OK;
}

int tbx_edit_device(tbx_engine* e, int op, const double* args, int n_args, int per_env, const uint8_t* mask_dev, void* stream)
{
    CHECK_ENGINE(e);
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, (hipStream_t)stream));
    TbxEditArgs a;
    int rc = edit_args(e, args, n_args, per_env, false, (hipStream_t)stream, a);
    if (rc) return rc;
    return e->ops->edit(e, op, a, mask_dev, (hipStream_t)stream);
}

int tbx_edit(tbx_engine* e, int op, const double* args, int n_args, int per_env, const uint8_t* mask_host)
{
    CHECK_ENGINE(e);
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    TbxEditArgs a;
    int rc = edit_args(e, args, n_args, per_env, true, e->stream, a);
    if (rc) return rc;
    const uint8_t* m = nullptr;
    if (mask_host) {
        EHIP(hipMemcpyAsync(e->mask, mask_host, (size_t)e->n, hipMemcpyHostToDevice, e->stream));
        m = e->mask;
    }
    rc = e->ops->edit(e, op, a, m, e->stream);
    if (rc) return rc;
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_reduce_device(tbx_engine* e, int query, const double* args, int n_args, int per_env, double* out_dev, void* stream)
{
    CHECK_ENGINE(e);
    const int width = tbx_reduce_width(e->game, query);
    if (width < 0) return e->fail(TBX_E_INVALID, "unknown query for this game");
    if (!out_dev) return e->fail(TBX_E_INVALID, "output pointer is NULL");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, (hipStream_t)stream));
    TbxEditArgs a;
    int rc = edit_args(e, args, n_args, per_env, false, (hipStream_t)stream, a);
    if (rc) return rc;
    return e->ops->reduce(e, query, a, out_dev, width, (hipStream_t)stream);
}

int tbx_reduce(tbx_engine* e, int query, const double* args, int n_args, int per_env, double* out_host)
{
    CHECK_ENGINE(e);
    const int width = tbx_reduce_width(e->game, query);
    if (width < 0) return e->fail(TBX_E_INVALID, "unknown query for this game");
    if (!out_host) return e->fail(TBX_E_INVALID, "output pointer is NULL");
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    TbxEditArgs a;
    int rc = edit_args(e, args, n_args, per_env, true, e->stream, a);
    if (rc) return rc;
    const size_t bytes = sizeof(double) * (size_t)e->n * (size_t)width;
    if (e->reduce_out_bytes < bytes) {
        EHIP(hipStreamSynchronize(e->stream));
        hipFree(e->reduce_out);
        e->reduce_out = nullptr; e->reduce_out_bytes = 0;
        EHIP(hipMalloc((void**)&e->reduce_out, bytes));
        e->reduce_out_bytes = bytes;
    }
    rc = e->ops->reduce(e, query, a, e->reduce_out, width, e->stream);
    if (rc) return rc;
    EHIP(hipMemcpyAsync(out_host, e->reduce_out, bytes, hipMemcpyDeviceToHost, e->stream));
    EHIP(hipStreamSynchronize(e->stream));
    return TBX_OK;
}

int tbx_query(tbx_engine* e, int env, int query_id, const int32_t* args, int n_args, int32_t* out, int n_out)
{
    CHECK_ENGINE(e);
    if (env < 0 || env >= e->n || !args || !out) return e->fail(TBX_E_INVALID, "bad query arguments");
    if (e->game == TBX_GAME_AMIDAR && n_args >= 2 && n_out >= 2) {
        // pure functions of the arguments (world = tile * (64, 80)); floor division for negative world coordinates
        if (query_id == TBX_QUERY_TILE_TO_WORLD) { out[0] = args[0] * TBX_AMI_TILE_WX; out[1] = args[1] * TBX_AMI_TILE_WY; return TBX_OK; }
        if (query_id == TBX_QUERY_WORLD_TO_TILE) {
            out[0] = args[0] >= 0 ? args[0] / TBX_AMI_TILE_WX : -((-args[0] + TBX_AMI_TILE_WX - 1) / TBX_AMI_TILE_WX);
            out[1] = args[1] >= 0 ? args[1] / TBX_AMI_TILE_WY : -((-args[1] + TBX_AMI_TILE_WY - 1) / TBX_AMI_TILE_WY);
            return TBX_OK;
        }
    }
    return e->fail(TBX_E_INVALID, "unknown query for this game");
}

int tbx_device_buffer(tbx_engine* e, int which, void** out_ptr, size_t* out_bytes)
{
    CHECK_ENGINE(e);
    if (!out_ptr) return e->fail(TBX_E_INVALID, "out_ptr is NULL");
    const size_t N = (size_t)e->n;
    void* p = nullptr;
    size_t b = 0;
    if ((which >= TBX_BUF_REWARD && which <= TBX_BUF_PACKED) || which == TBX_BUF_ROLLOUT_FRAMES || which == TBX_BUF_ROLLOUT_PACKED) {
        int rc = fused_reader_joins(e);                                // (overlapped fused launches, rollout chunks: the caller's stream joins here)
        if (!rc) rc = rollout_reader_joins(e, which == TBX_BUF_FRAME || which == TBX_BUF_ROLLOUT_FRAMES);
        if (rc) return rc;
    }
    switch (which) {
    case TBX_BUF_ROLLOUT_FRAMES:
        if (!e->pipe.chunk_k) return e->fail(TBX_E_INVALID, "tbx_rollout_synthetic has not been called");
        p = e->pipe.chunk_frames[e->pipe.chunk_cur];
        b = (size_t)e->pipe.chunk_k * N * e->ops->height() * e->ops->width() * e->pipe.chunk_channels;
        break;
    case TBX_BUF_ROLLOUT_PACKED:
        if (!e->pipe.chunk_k) return e->fail(TBX_E_INVALID, "tbx_rollout_synthetic has not been called");
        p = e->pipe.chunk_packed_base;
        b = sizeof(uint64_t) * (size_t)e->pipe.chunk_k * e->pipe.chunk_packed_stride;
        break;
    case TBX_BUF_REWARD: p = e->reward; b = N * 4; break;
    case TBX_BUF_DONE: p = e->done; b = N; break;
    case TBX_BUF_LIVES: p = e->lives_out; b = N * 4; break;
    case TBX_BUF_SCORE: p = e->score_out; b = N * 4; break;
    case TBX_BUF_FRAME: p = e->frame; b = e->frame_bytes; break;
    case TBX_BUF_PACKED: p = e->packed; b = N * 8; break;
    case TBX_BUF_AGENT_OBS: case TBX_BUF_AGENT_REWARD: case TBX_BUF_AGENT_DONE:
    case TBX_BUF_AGENT_EP_DONE: case TBX_BUF_AGENT_EP_RETURN: case TBX_BUF_AGENT_EP_LENGTH: case TBX_BUF_AGENT_PLANE: case TBX_BUF_AGENT_RING:
        return tbx_agent_buffer(e, which, out_ptr, out_bytes);
    case TBX_BUF_GATHERED: return tbx_gather_buffer(e, out_ptr, out_bytes);
    default: return e->fail(TBX_E_INVALID, "unknown buffer id");
    }
    *out_ptr = p;
    if (out_bytes) *out_bytes = b;
    return TBX_OK;
}

int tbx_set_option(tbx_engine* e, int option, int value)
{
    CHECK_ENGINE(e);
    bool ok = false;
    switch (option) {
    case TBX_OPT_PIPELINE: ok = value >= 0 && value <= 3; break;
    case TBX_OPT_STEP_FORM: ok = value >= 0 && value <= 2; break;
    case TBX_OPT_RENDER_SPLIT: ok = value >= 0 && value <= 64; break;
    case TBX_OPT_AGENT_GENERIC: case TBX_OPT_RESIDENT_STEP: case TBX_OPT_GATHER_TRANSPORT: ok = value == 0 || value == 1; break;
    case TBX_OPT_GATHER_EVERY: ok = value >= 1 && value <= 64; break;
    case TBX_OPT_FUSED_OVERLAP: ok = value >= 0 && value <= 2; break;
    case TBX_OPT_FUSED_OVERLAP_LEAD: ok = value >= 0 && value <= (1 << 20); break;
    case TBX_OPT_ROLLOUT_CHUNKS: ok = value >= 0 && value <= 4; break;
    default: return e->fail(TBX_E_INVALID, "unknown option");
    }
    if (!ok) return e->fail(TBX_E_INVALID, "option value out of range");
    if (e->opt[option] == value) return TBX_OK;
    // a launch-time choice must not change under work that is in flight
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_use_stream(e, e->stream));
    EHIP(hipStreamSynchronize(e->stream));
    e->opt[option] = value;
    e->ops->options_changed(e);
    return TBX_OK;
}

int tbx_get_option(tbx_engine* e, int option, int* value_out)
{
    CHECK_ENGINE(e);
    if (value_out && option == TBX_OPT_PIPELINE_ACTIVE) { *value_out = pipe_mode(e); return TBX_OK; }
    if (value_out && option == TBX_OPT_RECORDS_ACTIVE) { *value_out = e->ops->pipeline_ok() ? 1 : 0; return TBX_OK; }
    if (value_out && option == TBX_OPT_RENDER_STEP_FUSED) { *value_out = e->ops->render_step_fused(3) ? 1 : 0; return TBX_OK; }
    if (value_out && option == TBX_OPT_FUSED_OVERLAP_ACTIVE) { *value_out = fused_overlap_on(e, nullptr, 3) ? 1 : 0; return TBX_OK; }
    if (value_out && option == TBX_OPT_ROLLOUT_CHUNKS_ACTIVE) { *value_out = rollout_chunks_on(e, 3) ? 1 : 0; return TBX_OK; }
    if (option < 0 || option >= TBX_OPT_COUNT || !value_out) return e->fail(TBX_E_INVALID, "unknown option");
    *value_out = e->opt[option];
    return TBX_OK;
}

int tbx_sync(tbx_engine* e)
{
    CHECK_ENGINE(e);
    EHIP(hipSetDevice(e->device));
    EHIP(tbx_serve_stop(e));
    EHIP(tbx_finish_pending(e));
    EHIP(hipDeviceSynchronize());
    e->has_last = false;              // nothing is pending any more: the stream of the last call is no longer needed
    e->pipe.active = false;
    return check_err_flag(e);
}

}  // extern "C"
