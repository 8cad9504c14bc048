/* orc_gridworld.c -- CPU restatement of GridWorld.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * PARITY UNPINNED except for the data: the reference holds this game's default config and state dumps
 * (toybox/interventions/defaults/gridworld_config_default.json, gridworld_state_default.json) and the env class
 * (toybox/envs/atari/gridworld.py:8-13); the rules live in the absent ctoybox 0.5.0 core.  What the dumps pin:
 * the field set, the default board, tile attributes (color / goal / reward / walkable), player start and colour,
 * and that a state carries its own tile table with the grid as indices into it.  The movement rule, the reward
 * bookkeeping and the picture below are this repo's own specification (SPEC.md, "GridWorld"). */
#include "oracle.h"
#include <string.h>

void orc_gridworld_default_config(tbx_gridworld_config_t* c)
{
    static const char* rows[7] = {"111111111", "1000R0001", "101111101", "100010001", "10001R111", "1000100G1", "111111111"};
    static const char keys[4] = {'0', '1', 'G', 'R'};
    memset(c, 0, sizeof *c);
    c->width = 9; c->height = 7;
    c->n_tiles = 4;
    c->player_start_x = 2; c->player_start_y = 4;
    c->reward_becomes = 0;
    c->player_color = (tbx_color_t){255, 0, 0, 255};
    for (int i = 0; i < 4; i++) c->tile_keys[i] = (uint8_t)keys[i];
    c->tiles[0] = (tbx_gw_tile_t){{255, 255, 255, 255}, 0, 0, 1, {0, 0}};
    c->tiles[1] = (tbx_gw_tile_t){{0, 0, 0, 255}, 0, 0, 0, {0, 0}};
    c->tiles[2] = (tbx_gw_tile_t){{0, 255, 0, 255}, 10, 1, 1, {0, 0}};
    c->tiles[3] = (tbx_gw_tile_t){{255, 255, 0, 255}, 1, 0, 1, {0, 0}};
    for (int y = 0; y < 7; y++)
        for (int x = 0; x < 9; x++) {
            int id = 0;
            while (id < 4 && keys[id] != rows[y][x]) id++;
            c->grid[y * TBX_GW_MAX_DIM + x] = (uint8_t)id;
        }
}

/* a new game is the config's board with the player on its start cell; nothing is drawn from the RNG */
void orc_gridworld_new_game(const tbx_gridworld_config_t* c, uint64_t sim_rng[2], tbx_gridworld_state_t* s)
{
    (void)sim_rng;
    memset(s, 0, sizeof *s);
    s->player_x = c->player_start_x; s->player_y = c->player_start_y;
    s->reward_becomes = c->reward_becomes;
    s->width = c->width; s->height = c->height; s->n_tiles = c->n_tiles;
    s->player_color = c->player_color;
    memcpy(s->tiles, c->tiles, sizeof s->tiles);
    memcpy(s->grid, c->grid, sizeof s->grid);
}

/* one frame: at most one cell in the direction held (up, then down, then left, then right wins) */
void orc_gridworld_step(const tbx_gridworld_config_t* c, tbx_gridworld_state_t* s, uint32_t buttons)
{
    (void)c;
    if (s->game_over) return;
    int dx = 0, dy = 0;
    if (buttons & TBX_BTN_UP) dy = -1;
    else if (buttons & TBX_BTN_DOWN) dy = 1;
    else if (buttons & TBX_BTN_LEFT) dx = -1;
    else if (buttons & TBX_BTN_RIGHT) dx = 1;
    else return;
    const int nx = s->player_x + dx, ny = s->player_y + dy;
    if (nx < 0 || ny < 0 || nx >= s->width || ny >= s->height || nx >= TBX_GW_MAX_DIM || ny >= TBX_GW_MAX_DIM) return;
    const int id = s->grid[ny * TBX_GW_MAX_DIM + nx];
    if (id >= s->n_tiles || id >= TBX_GW_MAX_TILES) return;          /* a cell without a tile is a wall */
    const tbx_gw_tile_t* t = &s->tiles[id];
    if (!t->walkable) return;
    s->player_x = nx; s->player_y = ny;
    s->score += t->reward;
    if (t->reward != 0) s->grid[ny * TBX_GW_MAX_DIM + nx] = (uint8_t)s->reward_becomes;   /* collected */
    if (t->goal) s->game_over = 1;
}

static uint8_t gw_gray(tbx_color_t c) { return (uint8_t)((77u * c.r + 150u * c.g + 29u * c.b + 128u) >> 8); }

void orc_gridworld_render(const tbx_gridworld_config_t* c, const tbx_gridworld_state_t* s, uint8_t* out, int channels)
{
    (void)c;
    const int gw = s->width < 1 ? 1 : s->width > TBX_GW_MAX_DIM ? TBX_GW_MAX_DIM : s->width;
    const int gh = s->height < 1 ? 1 : s->height > TBX_GW_MAX_DIM ? TBX_GW_MAX_DIM : s->height;
    const int tw = TBX_GW_W / gw, th = TBX_GW_H / gh;
    for (int y = 0; y < TBX_GW_H; y++)
        for (int x = 0; x < TBX_GW_W; x++) {
            tbx_color_t col = {0, 0, 0, 255};
            const int cx = x / tw, cy = y / th;
            if (cx < gw && cy < gh) {
                if (cx == s->player_x && cy == s->player_y) col = s->player_color;
                else {
                    const int id = s->grid[cy * TBX_GW_MAX_DIM + cx];
                    if (id < s->n_tiles && id < TBX_GW_MAX_TILES) col = s->tiles[id].color;
                }
            }
            uint8_t* p = out + ((size_t)y * TBX_GW_W + x) * channels;
            if (channels == 1) p[0] = gw_gray(col);
            else { p[0] = col.r; p[1] = col.g; p[2] = col.b; if (channels == 4) p[3] = 255; }
        }
}
