/* selftest.c -- drives the CPU restatement through the C-ABI under AddressSanitizer + UBSan (`make -C oracle sanitize`).
 * TEST INFRASTRUCTURE ONLY.  GPU sanitizers are unavailable on this pool, so the checker itself is what gets sanitized:
 * rollouts of every game with auto-reset, all frame formats, state round trips, the agent layer with every wrapper on. */
#include "oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, tbx_last_error(e)); return 1; } } while (0)

static int run_game(int game, int n, int steps)
{
    tbx_engine* e = NULL;
    if (tbx_create(game, n, 0, NULL, 0, &e)) { fprintf(stderr, "create failed for game %d\n", game); return 1; }
    int h, w;
    CHECK(tbx_frame_dims(game, &h, &w));
    CHECK(tbx_seed(e, -1, 1234));
    CHECK(tbx_new_game(e, NULL));
    int32_t* actions = (int32_t*)malloc((size_t)n * 4);
    int32_t* reward = (int32_t*)malloc((size_t)n * 4);
    uint8_t* done = (uint8_t*)malloc((size_t)n);
    uint8_t* frame = (uint8_t*)malloc((size_t)n * h * w * 4);
    const size_t ssz = tbx_state_size(game);
    char* st = (char*)malloc(ssz * (size_t)n);
    long dones = 0;
    for (int t = 0; t < steps; t++) {
        for (int i = 0; i < n; i++) actions[i] = orc_synthetic_action(game, 1337, (uint64_t)i, (uint64_t)t);
        CHECK(tbx_step(e, actions, TBX_STEP_AUTO_RESET, reward, done, NULL, NULL));
        for (int i = 0; i < n; i++) dones += done[i];
        if (t % 97 == 0) {
            for (int c = 1; c <= 4; c += (c == 1 ? 2 : 1)) CHECK(tbx_render(e, frame, c));
            CHECK(tbx_get_states(e, 0, n, st, ssz));
            CHECK(tbx_set_states(e, 0, n, st, ssz));
        }
    }
    /* the agent layer with every wrapper on */
    tbx_agent_config_t ac;
    memset(&ac, 0, sizeof ac);
    ac.skip = 4; ac.out_h = 84; ac.out_w = 84; ac.stack = 4; ac.clip_reward = 1;
    ac.episodic_life = 1; ac.fire_reset = 1; ac.noop_max = 30; ac.noop_seed = 7; ac.env_offset = 11;
    CHECK(tbx_agent_init(e, &ac));
    uint8_t* obs = (uint8_t*)malloc((size_t)n * 84 * 84 * 4);
    float* ar = (float*)malloc((size_t)n * 4);
    CHECK(tbx_agent_reset(e, obs));
    for (int t = 0; t < steps / 4; t++) {
        for (int i = 0; i < n; i++) actions[i] = orc_synthetic_action(game, 99, (uint64_t)i, (uint64_t)t);
        CHECK(tbx_agent_step(e, actions, ar, done, obs));
        CHECK(tbx_agent_episodes(e, done, ar, reward));
    }
    printf("game %d: %d envs x %d frames ok (%ld episode ends)\n", game, n, steps, dones);
    free(actions); free(reward); free(done); free(frame); free(st); free(obs); free(ar);
    tbx_destroy(e);
    return 0;
}

int main(void)
{
    int rc = 0;
    rc |= run_game(TBX_GAME_BREAKOUT, 12, 2000);
    rc |= run_game(TBX_GAME_SPACE_INVADERS, 6, 2000);
    rc |= run_game(TBX_GAME_AMIDAR, 6, 2000);
    rc |= run_game(TBX_GAME_GRIDWORLD, 16, 1000);
    return rc;
}
