/* orc_breakout.c -- CPU restatement of Breakout.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Pinned by /root/reference/toybox/interventions/defaults/breakout_{config,state}_default.json
 * (tests/golden/breakout_initial.json): new_game() -- brick grid (index = col*rows+row, rect
 * (12+12c, 43+4r, 12, 4), depth = rows-1-row, points/colour by row), paddle (120,143) width 24
 * speed 4, ball radius 2, lives 5, is_dead = reset = true, one ball at a start position chosen by
 * orc_rng_range with velocity ball_speed_slow*(cos a, sin a), and the RNG bookkeeping.
 * Pinned by the reference's tests: FIRE leaves the paddle at (120,143)
 * (test/interventions/test_breakout_interventions.py:137-145) and keeps >=1 ball (:94-95).
 * Everything else in orc_breakout_step/_render is PARITY UNPINNED: it follows this repo's
 * specification (SPEC.md "Breakout"), modelled on toybox-rs' published behaviour.
 *
 * Arithmetic: IEEE-754 binary64, only + - * / sqrt ceil fabs compare; built with
 * -ffp-contract=off so no FMA is formed (the device build does the same). */
#include "oracle.h"
#include <math.h>
#include <string.h>

static tbx_color_t rgb(int r, int g, int b) { tbx_color_t c = {(uint8_t)r, (uint8_t)g, (uint8_t)b, 255}; return c; }

void orc_breakout_default_config(tbx_breakout_config_t* c)
{
    memset(c, 0, sizeof *c);
    orc_rng_seed(c->rand, 13);
    c->start_lives = 5;
    c->n_rows = 6;
    static const int scores[6] = {7, 7, 4, 4, 1, 1};
    static const int cols[6][3] = {{200, 72, 72}, {198, 108, 58}, {180, 122, 48},
                                   {162, 162, 42}, {72, 160, 72}, {66, 72, 200}};
    for (int i = 0; i < 6; i++) { c->row_scores[i] = scores[i]; c->row_colors[i] = rgb(cols[i][0], cols[i][1], cols[i][2]); }
    c->ball_speed_row_depth = 3;
    c->ball_speed_slow = 2.0;
    c->ball_speed_fast = 4.0;
    c->n_starts = 4;
    static const double sx[4] = {24.0, 120.0, 120.0, 216.0};
    static const double sa[4] = {30.0, 30.0, 150.0, 150.0};
    for (int i = 0; i < 4; i++) {
        c->start_x[i] = sx[i]; c->start_y[i] = 80.0; c->start_angle_deg[i] = sa[i];
        double rad = sa[i] * (M_PI / 180.0);
        c->start_dir_x[i] = cos(rad);
        c->start_dir_y[i] = sin(rad);
    }
    c->paddle_discrete_segments = 5;
    for (int i = 0; i < 5; i++) {
        double deg = 150.0 - (double)i * (120.0 / 4.0);
        double rad = deg * (M_PI / 180.0);
        c->paddle_dir_x[i] = cos(rad);
        c->paddle_dir_y[i] = -sin(rad);
    }
    c->bg_color = rgb(0, 0, 0);
    c->frame_color = rgb(144, 144, 144);
    c->paddle_color = rgb(200, 72, 72);
    c->ball_color = rgb(200, 72, 72);
}

static void start_ball(const tbx_breakout_config_t* c, tbx_breakout_state_t* s)
{
    uint64_t i = orc_rng_range(s->rand, (uint64_t)c->n_starts);
    int k = s->n_balls;
    if (k >= TBX_BRK_MAX_BALLS) return;
    s->ball_x[k] = c->start_x[i];
    s->ball_y[k] = c->start_y[i];
    s->ball_vx[k] = c->ball_speed_slow * c->start_dir_x[i];
    s->ball_vy[k] = c->ball_speed_slow * c->start_dir_y[i];
    s->n_balls = k + 1;
}

void orc_breakout_new_game(const tbx_breakout_config_t* c, uint64_t sim_rng[2], tbx_breakout_state_t* s)
{
    memset(s, 0, sizeof *s);
    orc_rng_child(sim_rng, s->rand);
    s->score = 0;
    s->lives = c->start_lives;
    s->level = 1;
    s->is_dead = 1;
    s->reset = 1;
    s->paddle_x = 120.0; s->paddle_y = 143.0; s->paddle_vx = 0.0; s->paddle_vy = 0.0;
    s->paddle_width = 24.0; s->paddle_speed = 4.0; s->ball_radius = 2.0;
    int rows = c->n_rows;
    s->n_bricks = TBX_BRK_COLS * rows;
    for (int col = 0; col < TBX_BRK_COLS; col++)
        for (int row = 0; row < rows; row++) {
            tbx_brick_t* b = &s->bricks[col * rows + row];
            b->x = TBX_BRK_LEFT + TBX_BRK_BRICK_W * (double)col;
            b->y = TBX_BRK_BRICK_Y0 + TBX_BRK_BRICK_H * (double)row;
            b->w = TBX_BRK_BRICK_W; b->h = TBX_BRK_BRICK_H;
            b->points = c->row_scores[row];
            b->depth = rows - 1 - row;
            b->row = row; b->col = col;
            b->color = c->row_colors[row];
            b->alive = 1; b->destructible = 1;
        }
    s->n_balls = 0;
    start_ball(c, s);
}

void orc_breakout_step(const tbx_breakout_config_t* c, tbx_breakout_state_t* s, uint32_t buttons)
{
    /* 1. paddle intent */
    if (buttons & TBX_BTN_LEFT) s->paddle_vx = -s->paddle_speed;
    else if (buttons & TBX_BTN_RIGHT) s->paddle_vx = s->paddle_speed;
    else s->paddle_vx = 0.0;
    s->paddle_vy = 0.0;

    /* 2. launch */
    if (s->is_dead && (buttons & TBX_BTN_BUTTON1)) { s->is_dead = 0; s->reset = 0; }
    int launched = !s->is_dead;

    /* 3. number of time slices: no ball moves further than its radius per slice */
    const double r = s->ball_radius;
    int nsl = 1;
    if (launched && r > 0.0) {
        double vmax = 0.0;
        for (int b = 0; b < s->n_balls; b++) {
            double m = sqrt(s->ball_vx[b] * s->ball_vx[b] + s->ball_vy[b] * s->ball_vy[b]);
            if (m > vmax) vmax = m;
        }
        double q = ceil(vmax / r);
        if (q > 16.0) q = 16.0;
        if (q >= 1.0) nsl = (int)q;
    }
    const double dt = 1.0 / (double)nsl;
    const double half = s->paddle_width * 0.5;
    int gone[TBX_BRK_MAX_BALLS] = {0, 0, 0, 0};

    for (int sl = 0; sl < nsl; sl++) {
        s->paddle_x = s->paddle_x + s->paddle_vx * dt;
        if (s->paddle_x - half < TBX_BRK_LEFT) s->paddle_x = TBX_BRK_LEFT + half;
        else if (s->paddle_x + half > TBX_BRK_RIGHT) s->paddle_x = TBX_BRK_RIGHT - half;
        if (!launched) continue;
        const double pl = s->paddle_x - half, pr = s->paddle_x + half;
        for (int b = 0; b < s->n_balls; b++) {
            if (gone[b]) continue;
            double x = s->ball_x[b] + s->ball_vx[b] * dt;
            double y = s->ball_y[b] + s->ball_vy[b] * dt;
            double vx = s->ball_vx[b], vy = s->ball_vy[b];
            /* walls */
            if (x - r < TBX_BRK_LEFT) vx = fabs(vx);
            if (x + r > TBX_BRK_RIGHT) vx = -fabs(vx);
            if (y - r < TBX_BRK_TOP) vy = fabs(vy);
            /* paddle */
            if (vy > 0.0 && y + r >= s->paddle_y && y - r <= s->paddle_y + TBX_BRK_PADDLE_H &&
                x + r >= pl && x - r <= pr) {
                int S = c->paddle_discrete_segments;
                double t = (x - pl) / s->paddle_width;
                if (t < 0.0) t = 0.0;
                if (t > 1.0) t = 1.0;
                int seg = (int)(t * (double)S);
                if (seg > S - 1) seg = S - 1;
                double sp = sqrt(vx * vx + vy * vy);
                vx = sp * c->paddle_dir_x[seg];
                vy = sp * c->paddle_dir_y[seg];
            }
            /* bricks: lowest-index alive brick whose rect overlaps the ball's box */
            int hit = -1;
            for (int i = 0; i < s->n_bricks; i++) {
                const tbx_brick_t* k = &s->bricks[i];
                if (!k->alive) continue;
                if (x + r > k->x && x - r < k->x + k->w && y + r > k->y && y - r < k->y + k->h) { hit = i; break; }
            }
            if (hit >= 0) {
                tbx_brick_t* k = &s->bricks[hit];
                int cx_in = (x >= k->x && x <= k->x + k->w);
                int cy_in = (y >= k->y && y <= k->y + k->h);
                if (cy_in && !cx_in) vx = (x < k->x) ? -fabs(vx) : fabs(vx);
                else vy = (y < k->y + k->h * 0.5) ? -fabs(vy) : fabs(vy);
                if (k->destructible) { k->alive = 0; s->score += k->points; }
                if (k->depth >= c->ball_speed_row_depth) {
                    double m = sqrt(vx * vx + vy * vy);
                    if (m < c->ball_speed_fast && m > 0.0) {
                        double f = c->ball_speed_fast / m;
                        vx = vx * f; vy = vy * f;
                    }
                }
            }
            if (y - r > TBX_BRK_BOTTOM) gone[b] = 1;
            s->ball_x[b] = x; s->ball_y[b] = y; s->ball_vx[b] = vx; s->ball_vy[b] = vy;
        }
    }

    /* 4. remove lost balls (order kept), life lost when none remain */
    if (launched) {
        int k = 0;
        for (int b = 0; b < s->n_balls; b++) {
            if (gone[b]) continue;
            s->ball_x[k] = s->ball_x[b]; s->ball_y[k] = s->ball_y[b];
            s->ball_vx[k] = s->ball_vx[b]; s->ball_vy[k] = s->ball_vy[b];
            k++;
        }
        for (int b = k; b < TBX_BRK_MAX_BALLS; b++) { s->ball_x[b] = s->ball_y[b] = s->ball_vx[b] = s->ball_vy[b] = 0.0; }
        s->n_balls = k;
        if (k == 0) {
            s->lives -= 1;
            s->is_dead = 1; s->reset = 1;
            start_ball(c, s);
        }
    }

    /* 5. wall cleared -> next level */
    int n_d = 0, n_alive = 0;
    for (int i = 0; i < s->n_bricks; i++)
        if (s->bricks[i].destructible) { n_d++; n_alive += s->bricks[i].alive != 0; }
    if (n_d > 0 && n_alive == 0) {
        s->level += 1;
        for (int i = 0; i < s->n_bricks; i++) if (s->bricks[i].destructible) s->bricks[i].alive = 1;
    }
}

/* ---------------------------------------------------------------- render */

/* 3x5 digit font, bit (row*3+col), col 0 = left */
static const uint16_t DIGITS[10] = {
    /* 0 */ 0x7B6F, /* 1 */ 0x749A, /* 2 */ 0x73E7, /* 3 */ 0x79E7, /* 4 */ 0x49ED,
    /* 5 */ 0x79CF, /* 6 */ 0x7BCF, /* 7 */ 0x4927, /* 8 */ 0x7BEF, /* 9 */ 0x79EF,
};

static int f2i(double v)
{
    if (!(v > -1.0e6)) v = -1.0e6;
    if (v > 1.0e6) v = 1.0e6;
    return (int)v;
}

static void put(uint8_t* out, int ch, int x, int y, tbx_color_t c)
{
    if (x < 0 || y < 0 || x >= TBX_BRK_W || y >= TBX_BRK_H) return;
    uint8_t* p = out + ((size_t)y * TBX_BRK_W + x) * ch;
    if (ch == 1) p[0] = (uint8_t)((77 * c.r + 150 * c.g + 29 * c.b + 128) >> 8);
    else { p[0] = c.r; p[1] = c.g; p[2] = c.b; if (ch == 4) p[3] = 255; }
}

static void rect(uint8_t* out, int ch, int x0, int y0, int w, int h, tbx_color_t c)
{
    long xa = x0, xb = (long)x0 + w, ya = y0, yb = (long)y0 + h;   /* clip first: sizes come from state records */
    if (xa < 0) xa = 0;
    if (ya < 0) ya = 0;
    if (xb > TBX_BRK_W) xb = TBX_BRK_W;
    if (yb > TBX_BRK_H) yb = TBX_BRK_H;
    for (long y = ya; y < yb; y++)
        for (long x = xa; x < xb; x++) put(out, ch, (int)x, (int)y, c);
}

static void digit(uint8_t* out, int ch, int x0, int y0, int d, tbx_color_t c)
{
    uint16_t g = DIGITS[d];
    for (int py = 0; py < 10; py++)
        for (int px = 0; px < 6; px++)
            if ((g >> ((py / 2) * 3 + (px / 2))) & 1) put(out, ch, x0 + px, y0 + py, c);
}

void orc_breakout_render(const tbx_breakout_config_t* c, const tbx_breakout_state_t* s, uint8_t* out, int ch)
{
    rect(out, ch, 0, 0, TBX_BRK_W, TBX_BRK_H, c->bg_color);
    /* frame: top bar and the two side walls */
    rect(out, ch, 0, TBX_BRK_WALL_Y0, TBX_BRK_W, 12, c->frame_color);
    rect(out, ch, 0, TBX_BRK_WALL_Y0, 12, TBX_BRK_H - TBX_BRK_WALL_Y0, c->frame_color);
    rect(out, ch, 228, TBX_BRK_WALL_Y0, 12, TBX_BRK_H - TBX_BRK_WALL_Y0, c->frame_color);
    /* bricks in index order */
    for (int i = 0; i < s->n_bricks; i++) {
        const tbx_brick_t* k = &s->bricks[i];
        if (k->alive) rect(out, ch, f2i(k->x), f2i(k->y), f2i(k->w), f2i(k->h), k->color);
    }
    /* paddle, balls */
    rect(out, ch, f2i(s->paddle_x - s->paddle_width * 0.5), f2i(s->paddle_y), f2i(s->paddle_width), 3, c->paddle_color);
    for (int b = 0; b < s->n_balls; b++)
        rect(out, ch, f2i(s->ball_x[b] - s->ball_radius), f2i(s->ball_y[b] - s->ball_radius),
             f2i(s->ball_radius * 2.0), f2i(s->ball_radius * 2.0), c->ball_color);
    /* HUD: score (5 digits, zero padded), lives, level */
    int sc = s->score; if (sc < 0) sc = 0; sc %= 100000;
    int div = 10000;
    for (int i = 0; i < 5; i++) { digit(out, ch, 36 + 8 * i, 2, (sc / div) % 10, c->frame_color); div /= 10; }
    int lv = s->lives; if (lv < 0) lv = 0; if (lv > 9) lv = 9;
    digit(out, ch, 148, 2, lv, c->frame_color);
    int le = s->level; if (le < 0) le = 0;
    digit(out, ch, 196, 2, le % 10, c->frame_color);
}
