/* orc_amidar.c -- CPU restatement of Amidar.  TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Pinned by /root/reference/toybox/interventions/defaults/amidar_{config,state}_default.json
 * (tests/golden/amidar_state.json): new_game() -- tiles from the config board, the 29 boxes
 * (as a set) with triggers_chase on the chase-marker corners, chase_junctions [0,25,768,793],
 * player at tile (31,15) = world (1984,1200) with history [607], five EnemyLookupAI enemies of
 * speed 8 at worlds (0,0) (0,0) (448,0) (0,2000) (576,2400), lives 3, jumps 4, timers 0, RNG (KAT-A).
 * Pinned by the reference's tests: one FIRE leaves jumps == 3
 * (test/interventions/test_amidar_interventions.py:170-173); tile<->world queries scale by (64,80).
 * Everything else (movement, painting, enemy protocols, default routes, pixels) is PARITY
 * UNPINNED and follows SPEC.md "Amidar".  Integer arithmetic only. */
#include "oracle.h"
#include "../include/toybox_amd_spec.h"
#include <string.h>

#define BW TBX_AMI_BOARD_W
#define BH TBX_AMI_BOARD_H

static const int ROUTES[TBX_AMI_N_ROUTES][TBX_AMI_ROUTE_LEN] = TBX_AMI_ROUTES;

static const char* DEFAULT_BOARD[BH] = {
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

static tbx_color_t rgb4(int r, int g, int b) { tbx_color_t c = {(uint8_t)r, (uint8_t)g, (uint8_t)b, 255}; return c; }

void orc_amidar_default_config(tbx_amidar_config_t* c)
{
    memset(c, 0, sizeof *c);
    orc_rng_seed(c->rand, 13);
    c->start_lives = 3; c->start_jumps = 4; c->jump_time = 75; c->chase_time = 300;
    c->box_bonus = 50; c->chase_score_bonus = 100;
    c->player_start_tx = 31; c->player_start_ty = 15;
    c->n_enemies = 5;
    c->render_images = 1; c->default_board_bugs = 1;
    for (int i = 0; i < 5; i++) {
        tbx_amidar_ai_t* a = &c->enemies[i];
        memset(a, 0, sizeof *a);
        a->kind = TBX_AI_LOOKUP; a->next = 0; a->default_route_index = i;
        a->seen_tx = a->seen_ty = -1;
    }
    c->bg_color = rgb4(0, 0, 0); c->player_color = rgb4(255, 255, 153); c->unpainted_color = rgb4(148, 0, 211);
    c->painted_color = rgb4(255, 255, 30); c->enemy_color = rgb4(255, 50, 100); c->inner_painted_color = rgb4(255, 255, 0);
    for (int y = 0; y < BH; y++)
        for (int x = 0; x < BW; x++) {
            char ch = DEFAULT_BOARD[y][x];
            c->board[y][x] = ch == '=' ? TBX_TILE_UNPAINTED : ch == 'p' ? TBX_TILE_PAINTED : ch == 'c' ? TBX_TILE_CHASE_MARKER : TBX_TILE_EMPTY;
        }
}

/* ---------------------------------------------------------------- board helpers */

static int walkable(const tbx_amidar_state_t* s, int tx, int ty)
{
    return tx >= 0 && ty >= 0 && tx < BW && ty < BH && s->tiles[ty][tx] != TBX_TILE_EMPTY;
}

static int is_junction(const tbx_amidar_state_t* s, int tx, int ty)
{
    if (!walkable(s, tx, ty)) return 0;
    int h = walkable(s, tx - 1, ty) || walkable(s, tx + 1, ty);
    int v = walkable(s, tx, ty - 1) || walkable(s, tx, ty + 1);
    return h && v;
}

static void dir_delta(int dir, int* dx, int* dy)
{
    *dx = dir == TBX_DIR_LEFT ? -1 : dir == TBX_DIR_RIGHT ? 1 : 0;
    *dy = dir == TBX_DIR_UP ? -1 : dir == TBX_DIR_DOWN ? 1 : 0;
}

static int opposite(int dir) { return dir ^ 1; }   /* Up<->Down, Left<->Right */

static int can_go(const tbx_amidar_state_t* s, int tx, int ty, int dir)
{
    int dx, dy;
    dir_delta(dir, &dx, &dy);
    return walkable(s, tx + dx, ty + dy);
}

/* boxes = maximal empty rectangles, recorded by their track corners, scanned row-major */
static void find_boxes(tbx_amidar_state_t* s)
{
    s->n_boxes = 0;
    for (int ty = 0; ty < BH - 1; ty++)
        for (int tx = 0; tx < BW - 1; tx++) {
            /* (tx,ty) is a top-left corner when the tile right-below is empty and its up/left neighbours are track */
            if (!walkable(s, tx, ty) || !walkable(s, tx + 1, ty) || !walkable(s, tx, ty + 1) || walkable(s, tx + 1, ty + 1)) continue;
            int x1 = tx + 1, y1 = ty + 1;
            while (x1 < BW && !walkable(s, x1, ty + 1)) x1++;
            while (y1 < BH && !walkable(s, tx + 1, y1)) y1++;
            if (x1 >= BW || y1 >= BH || s->n_boxes >= TBX_AMI_MAX_BOXES) continue;
            tbx_amidar_box_t* b = &s->boxes[s->n_boxes++];
            b->tl_tx = tx; b->tl_ty = ty; b->br_tx = x1; b->br_ty = y1;
            b->painted = 0;
            b->triggers_chase = 0;
            for (int k = 0; k < s->n_chase_junctions; k++)
                if (s->chase_junctions[k] == ty * BW + tx) b->triggers_chase = 1;
        }
}

static void reset_mover(tbx_amidar_mover_t* m, int tx, int ty)
{
    m->x = tx * TBX_AMI_TILE_WX; m->y = ty * TBX_AMI_TILE_WY;
    m->step_tx = m->step_ty = -1;
    m->n_history = 0;
    memset(m->history, 0, sizeof m->history);
    m->caught = 0;
}

static void enemy_start_tile(const tbx_amidar_ai_t* a, int* tx, int* ty)
{
    if (a->kind == TBX_AI_LOOKUP) {
        int r = a->default_route_index;
        int id = (r >= 0 && r < TBX_AMI_N_ROUTES) ? ROUTES[r][0] : 0;
        *tx = id % BW; *ty = id / BW;
    } else { *tx = a->start_tx; *ty = a->start_ty; }
}

static void reset_enemy(tbx_amidar_mover_t* m)
{
    int tx, ty;
    tbx_amidar_ai_t* a = &m->ai;
    if (a->kind == TBX_AI_LOOKUP) a->next = 0;
    a->vert = a->start_vert; a->horiz = a->start_horiz; a->dir = a->start_dir;
    a->seen_tx = a->seen_ty = -1;
    enemy_start_tile(a, &tx, &ty);
    reset_mover(m, tx, ty);
}

static void reset_player(const tbx_amidar_config_t* c, tbx_amidar_state_t* s)
{
    tbx_amidar_mover_t* p = &s->player;
    reset_mover(p, c->player_start_tx, c->player_start_ty);
    /* history starts with the first junction found below the start tile (607 for the golden board) */
    for (int ty = c->player_start_ty + 1; ty < BH; ty++)
        if (is_junction(s, c->player_start_tx, ty)) { p->history[0] = ty * BW + c->player_start_tx; p->n_history = 1; break; }
}

static void reset_board(const tbx_amidar_config_t* c, tbx_amidar_state_t* s)
{
    memcpy(s->tiles, c->board, sizeof s->tiles);
    s->n_chase_junctions = 0;
    for (int ty = 0; ty < BH; ty++)
        for (int tx = 0; tx < BW; tx++)
            if (s->tiles[ty][tx] == TBX_TILE_CHASE_MARKER && s->n_chase_junctions < TBX_AMI_MAX_CHASE_J)
                s->chase_junctions[s->n_chase_junctions++] = ty * BW + tx;
    memset(s->boxes, 0, sizeof s->boxes);
    find_boxes(s);
}

void orc_amidar_new_game(const tbx_amidar_config_t* c, uint64_t sim_rng[2], tbx_amidar_state_t* s)
{
    memset(s, 0, sizeof *s);
    orc_rng_child(sim_rng, s->rand);
    s->score = 0; s->lives = c->start_lives; s->level = 1;
    s->jumps = c->start_jumps; s->jump_timer = 0; s->chase_timer = 0;
    reset_board(c, s);
    s->player.speed = TBX_AMI_SPEED;
    s->player.ai.kind = TBX_AI_PLAYER;
    s->player.ai.seen_tx = s->player.ai.seen_ty = -1;
    reset_player(c, s);
    s->n_enemies = c->n_enemies;
    for (int i = 0; i < c->n_enemies; i++) {
        tbx_amidar_mover_t* m = &s->enemies[i];
        m->speed = TBX_AMI_SPEED;
        m->ai = c->enemies[i];
        reset_enemy(m);
    }
}

/* ---------------------------------------------------------------- movement */

static int at_tile(const tbx_amidar_mover_t* m) { return m->x % TBX_AMI_TILE_WX == 0 && m->y % TBX_AMI_TILE_WY == 0; }

/* advance toward step; returns 1 when the target tile was reached this frame */
static int advance(tbx_amidar_mover_t* m)
{
    if (m->step_tx < 0) return 0;
    int gx = m->step_tx * TBX_AMI_TILE_WX, gy = m->step_ty * TBX_AMI_TILE_WY;
    int sp = m->speed < 0 ? 0 : m->speed;
    if (m->x < gx) { m->x += sp; if (m->x > gx) m->x = gx; }
    else if (m->x > gx) { m->x -= sp; if (m->x < gx) m->x = gx; }
    else if (m->y < gy) { m->y += sp; if (m->y > gy) m->y = gy; }
    else if (m->y > gy) { m->y -= sp; if (m->y < gy) m->y = gy; }
    if (m->x == gx && m->y == gy) { m->step_tx = m->step_ty = -1; return 1; }
    return 0;
}

static void set_step(tbx_amidar_mover_t* m, int tx, int ty, int dir)
{
    int dx, dy;
    dir_delta(dir, &dx, &dy);
    m->step_tx = tx + dx; m->step_ty = ty + dy;
}

static void push_history(tbx_amidar_mover_t* m, int id)
{
    if (m->n_history >= TBX_AMI_MAX_HISTORY) {
        for (int i = 1; i < TBX_AMI_MAX_HISTORY; i++) m->history[i - 1] = m->history[i];
        m->n_history = TBX_AMI_MAX_HISTORY - 1;
    }
    m->history[m->n_history++] = id;
}

static void check_boxes(const tbx_amidar_config_t* c, tbx_amidar_state_t* s)
{
    for (int i = 0; i < s->n_boxes; i++) {
        tbx_amidar_box_t* b = &s->boxes[i];
        if (b->painted) continue;
        int ok = 1;
        for (int x = b->tl_tx; x <= b->br_tx && ok; x++)
            if (s->tiles[b->tl_ty][x] != TBX_TILE_PAINTED || s->tiles[b->br_ty][x] != TBX_TILE_PAINTED) ok = 0;
        for (int y = b->tl_ty; y <= b->br_ty && ok; y++)
            if (s->tiles[y][b->tl_tx] != TBX_TILE_PAINTED || s->tiles[y][b->br_tx] != TBX_TILE_PAINTED) ok = 0;
        if (!ok) continue;
        b->painted = 1;
        s->score += c->box_bonus;
        if (b->triggers_chase) {
            int all = 1;
            for (int k = 0; k < s->n_boxes; k++) if (s->boxes[k].triggers_chase && !s->boxes[k].painted) all = 0;
            if (all) s->chase_timer = c->chase_time;
        }
    }
}

static void reset_positions(const tbx_amidar_config_t* c, tbx_amidar_state_t* s)
{
    reset_player(c, s);
    for (int i = 0; i < s->n_enemies; i++) reset_enemy(&s->enemies[i]);
    s->jump_timer = 0; s->chase_timer = 0;
}

static void player_arrived(const tbx_amidar_config_t* c, tbx_amidar_state_t* s)
{
    tbx_amidar_mover_t* p = &s->player;
    int tx = p->x / TBX_AMI_TILE_WX, ty = p->y / TBX_AMI_TILE_WY;
    if (!is_junction(s, tx, ty)) return;
    int id = ty * BW + tx, newly = 0;
    if (p->n_history > 0) {
        int prev = p->history[p->n_history - 1];
        int px = prev % BW, py = prev / BW;
        if (prev != id && prev >= 0 && prev < BW * BH && (px == tx || py == ty)) {
            int x0 = px < tx ? px : tx, x1 = px < tx ? tx : px, y0 = py < ty ? py : ty, y1 = py < ty ? ty : py;
            int clear = 1;
            for (int y = y0; y <= y1; y++) for (int x = x0; x <= x1; x++) if (!walkable(s, x, y)) clear = 0;
            if (clear)
                for (int y = y0; y <= y1; y++) for (int x = x0; x <= x1; x++)
                    if (s->tiles[y][x] != TBX_TILE_PAINTED) { s->tiles[y][x] = TBX_TILE_PAINTED; newly++; }
        }
    }
    push_history(p, id);
    if (newly > 0) {
        s->score += newly;
        check_boxes(c, s);
        int left = 0;
        for (int y = 0; y < BH; y++) for (int x = 0; x < BW; x++)
            if (s->tiles[y][x] == TBX_TILE_UNPAINTED || s->tiles[y][x] == TBX_TILE_CHASE_MARKER) left = 1;
        if (!left) {   /* board complete */
            s->level += 1;
            reset_board(c, s);
            reset_positions(c, s);
            s->jumps = c->start_jumps;
        }
    }
}

static int first_open(const tbx_amidar_state_t* s, int tx, int ty, int avoid)
{
    for (int d = 0; d < 4; d++) if (d != avoid && can_go(s, tx, ty, d)) return d;
    return avoid >= 0 && can_go(s, tx, ty, avoid) ? avoid : -1;
}

static void enemy_decide(tbx_amidar_state_t* s, tbx_amidar_mover_t* m)
{
    tbx_amidar_ai_t* a = &m->ai;
    int tx = m->x / TBX_AMI_TILE_WX, ty = m->y / TBX_AMI_TILE_WY;
    int dir = -1;
    switch (a->kind) {
    case TBX_AI_LOOKUP: {
        int r = a->default_route_index;
        if (r < 0 || r >= TBX_AMI_N_ROUTES) return;   /* no such table: the enemy idles */
        int len = 0;
        while (len < TBX_AMI_ROUTE_LEN && ROUTES[r][len] >= 0) len++;
        if (a->next < 0 || a->next >= len) a->next = 0;
        if (ROUTES[r][a->next] == ty * BW + tx) a->next = (a->next + 1) % len;
        int gx = ROUTES[r][a->next] % BW, gy = ROUTES[r][a->next] / BW;
        if (gx > tx && can_go(s, tx, ty, TBX_DIR_RIGHT)) dir = TBX_DIR_RIGHT;
        else if (gx < tx && can_go(s, tx, ty, TBX_DIR_LEFT)) dir = TBX_DIR_LEFT;
        else if (gy > ty && can_go(s, tx, ty, TBX_DIR_DOWN)) dir = TBX_DIR_DOWN;
        else if (gy < ty && can_go(s, tx, ty, TBX_DIR_UP)) dir = TBX_DIR_UP;
        break; }
    case TBX_AI_PERIMETER:
        if (ty == 0 && tx < BW - 1 && can_go(s, tx, ty, TBX_DIR_RIGHT)) dir = TBX_DIR_RIGHT;
        else if (tx == BW - 1 && ty < BH - 1 && can_go(s, tx, ty, TBX_DIR_DOWN)) dir = TBX_DIR_DOWN;
        else if (ty == BH - 1 && tx > 0 && can_go(s, tx, ty, TBX_DIR_LEFT)) dir = TBX_DIR_LEFT;
        else if (tx == 0 && ty > 0 && can_go(s, tx, ty, TBX_DIR_UP)) dir = TBX_DIR_UP;
        else {
            static const int order[4] = {TBX_DIR_UP, TBX_DIR_LEFT, TBX_DIR_DOWN, TBX_DIR_RIGHT};
            for (int k = 0; k < 4 && dir < 0; k++) if (can_go(s, tx, ty, order[k])) dir = order[k];
        }
        break;
    case TBX_AI_AMIDAR: {
        a->vert &= 1;                       /* Up / Down */
        a->horiz = 2 | (a->horiz & 1);      /* Left / Right */
        int came_vertically = 1;
        if (m->n_history > 0) came_vertically = (m->history[m->n_history - 1] % BW) == tx;
        int can_h = can_go(s, tx, ty, a->horiz), can_v = can_go(s, tx, ty, a->vert);
        if (came_vertically && can_h) dir = a->horiz;
        else if (can_v) dir = a->vert;
        else if (can_h) dir = a->horiz;
        else {
            a->vert = opposite(a->vert);
            a->horiz = opposite(a->horiz);
            if (can_go(s, tx, ty, a->vert)) dir = a->vert;
            else if (can_go(s, tx, ty, a->horiz)) dir = a->horiz;
        }
        break; }
    case TBX_AI_TARGET_PLAYER: {
        int ptx = s->player.x / TBX_AMI_TILE_WX, pty = s->player.y / TBX_AMI_TILE_WY;
        int ddx = ptx - tx, ddy = pty - ty;
        int adx = ddx < 0 ? -ddx : ddx, ady = ddy < 0 ? -ddy : ddy;
        a->dir &= 3;
        if (adx + ady <= a->vision_distance) {
            a->seen_tx = ptx; a->seen_ty = pty;
            int hd = ddx > 0 ? TBX_DIR_RIGHT : TBX_DIR_LEFT, vd = ddy > 0 ? TBX_DIR_DOWN : TBX_DIR_UP;
            int first = adx >= ady ? hd : vd, second = adx >= ady ? vd : hd;
            int fz = adx >= ady ? adx : ady, sz = adx >= ady ? ady : adx;
            if (fz > 0 && can_go(s, tx, ty, first)) dir = first;
            else if (sz > 0 && can_go(s, tx, ty, second)) dir = second;
        } else { a->seen_tx = a->seen_ty = -1; }
        if (dir < 0) dir = can_go(s, tx, ty, a->dir) ? a->dir : first_open(s, tx, ty, opposite(a->dir));
        if (dir >= 0) a->dir = dir;
        break; }
    case TBX_AI_RANDOM: {
        a->dir &= 3;
        int opts[4], n = 0;
        for (int d = 0; d < 4; d++) if (d != opposite(a->dir) && can_go(s, tx, ty, d)) opts[n++] = d;
        if (n == 0) dir = can_go(s, tx, ty, opposite(a->dir)) ? opposite(a->dir) : -1;
        else dir = opts[orc_rng_range(s->rand, (uint64_t)n)];
        if (dir >= 0) a->dir = dir;
        break; }
    default: return;
    }
    if (dir >= 0) set_step(m, tx, ty, dir);
}

void orc_amidar_step(const tbx_amidar_config_t* c, tbx_amidar_state_t* s, uint32_t buttons)
{
    /* 1. timers */
    if (s->jump_timer > 0) s->jump_timer -= 1;
    if (s->chase_timer > 0) {
        s->chase_timer -= 1;
        if (s->chase_timer == 0)
            for (int i = 0; i < s->n_enemies; i++) if (s->enemies[i].caught) reset_enemy(&s->enemies[i]);
    }
    /* 2. jump */
    if ((buttons & TBX_BTN_BUTTON1) && s->jumps > 0 && s->jump_timer == 0) { s->jumps -= 1; s->jump_timer = c->jump_time; }

    /* 3. player */
    tbx_amidar_mover_t* p = &s->player;
    if (p->step_tx < 0 && at_tile(p)) {
        int tx = p->x / TBX_AMI_TILE_WX, ty = p->y / TBX_AMI_TILE_WY;
        int dir = (buttons & TBX_BTN_UP) ? TBX_DIR_UP : (buttons & TBX_BTN_DOWN) ? TBX_DIR_DOWN :
                  (buttons & TBX_BTN_LEFT) ? TBX_DIR_LEFT : (buttons & TBX_BTN_RIGHT) ? TBX_DIR_RIGHT : -1;
        if (dir >= 0 && can_go(s, tx, ty, dir)) set_step(p, tx, ty, dir);
    }
    const int level_before = s->level;
    if (advance(p)) player_arrived(c, s);
    if (s->level != level_before) return;   /* board complete: everything was re-placed */

    /* 4. enemies, in index order */
    for (int i = 0; i < s->n_enemies; i++) {
        tbx_amidar_mover_t* m = &s->enemies[i];
        if (m->caught) continue;
        if (m->step_tx < 0 && at_tile(m)) enemy_decide(s, m);
        if (advance(m)) {
            int id = (m->y / TBX_AMI_TILE_WY) * BW + m->x / TBX_AMI_TILE_WX;
            if (m->ai.kind != TBX_AI_LOOKUP) { m->n_history = 0; push_history(m, id); }
        }
    }

    /* 5. collisions */
    for (int i = 0; i < s->n_enemies; i++) {
        tbx_amidar_mover_t* m = &s->enemies[i];
        if (m->caught) continue;
        int dx = m->x - p->x, dy = m->y - p->y;
        if (dx < 0) dx = -dx;
        if (dy < 0) dy = -dy;
        if (dx >= TBX_AMI_HIT_DX || dy >= TBX_AMI_HIT_DY) continue;
        if (s->jump_timer > 0) continue;
        if (s->chase_timer > 0) { m->caught = 1; s->score += c->chase_score_bonus; continue; }
        s->lives -= 1;
        reset_positions(c, s);
        break;
    }
}

/* ---------------------------------------------------------------- queries */

int orc_amidar_query(int query_id, const int32_t* args, int32_t* out)
{
    if (query_id == TBX_QUERY_TILE_TO_WORLD) { out[0] = args[0] * TBX_AMI_TILE_WX; out[1] = args[1] * TBX_AMI_TILE_WY; return 0; }
    if (query_id == TBX_QUERY_WORLD_TO_TILE) {
        /* floor division: world coordinates may be negative */
        out[0] = args[0] >= 0 ? args[0] / TBX_AMI_TILE_WX : -((-args[0] + TBX_AMI_TILE_WX - 1) / TBX_AMI_TILE_WX);
        out[1] = args[1] >= 0 ? args[1] / TBX_AMI_TILE_WY : -((-args[1] + TBX_AMI_TILE_WY - 1) / TBX_AMI_TILE_WY);
        return 0;
    }
    return -1;
}

/* ---------------------------------------------------------------- render */

static const uint16_t DIGITS[10] = TBX_DIGIT_FONT;

static void put(uint8_t* out, int ch, int x, int y, tbx_color_t c)
{
    if (x < 0 || y < 0 || x >= TBX_AMI_W || y >= TBX_AMI_H) return;
    uint8_t* p = out + ((size_t)y * TBX_AMI_W + x) * ch;
    if (ch == 1) p[0] = (uint8_t)((77 * c.r + 150 * c.g + 29 * c.b + 128) >> 8);
    else { p[0] = c.r; p[1] = c.g; p[2] = c.b; if (ch == 4) p[3] = 255; }
}

static void rect(uint8_t* out, int ch, int x0, int y0, int w, int h, tbx_color_t c)
{
    long xa = x0, xb = (long)x0 + w, ya = y0, yb = (long)y0 + h;
    if (xa < 0) xa = 0;
    if (ya < 0) ya = 0;
    if (xb > TBX_AMI_W) xb = TBX_AMI_W;
    if (yb > TBX_AMI_H) yb = TBX_AMI_H;
    for (long y = ya; y < yb; y++)
        for (long x = xa; x < xb; x++) put(out, ch, (int)x, (int)y, c);
}

static void digit(uint8_t* out, int ch, int x0, int y0, int d, tbx_color_t c)
{
    for (int py = 0; py < 10; py++)
        for (int px = 0; px < 6; px++)
            if ((DIGITS[d] >> ((py / 2) * 3 + (px / 2))) & 1) put(out, ch, x0 + px, y0 + py, c);
}

static int world_to_px(int v)   /* floor(v / 16) */
{
    return v >= 0 ? v / TBX_AMI_WORLD_SCALE : -((-v + TBX_AMI_WORLD_SCALE - 1) / TBX_AMI_WORLD_SCALE);
}

void orc_amidar_render(const tbx_amidar_config_t* c, const tbx_amidar_state_t* s, uint8_t* out, int ch)
{
    rect(out, ch, 0, 0, TBX_AMI_W, TBX_AMI_H, c->bg_color);
    /* painted boxes' interiors */
    for (int i = 0; i < s->n_boxes; i++) {
        const tbx_amidar_box_t* b = &s->boxes[i];
        if (!b->painted) continue;
        rect(out, ch, TBX_AMI_BOARD_OX + TBX_AMI_TILE_PW * (b->tl_tx + 1), TBX_AMI_BOARD_OY + TBX_AMI_TILE_PH * (b->tl_ty + 1),
             TBX_AMI_TILE_PW * (b->br_tx - b->tl_tx - 1), TBX_AMI_TILE_PH * (b->br_ty - b->tl_ty - 1), c->inner_painted_color);
    }
    /* track */
    for (int ty = 0; ty < BH; ty++)
        for (int tx = 0; tx < BW; tx++) {
            int t = s->tiles[ty][tx];
            if (t == TBX_TILE_EMPTY) continue;
            rect(out, ch, TBX_AMI_BOARD_OX + TBX_AMI_TILE_PW * tx, TBX_AMI_BOARD_OY + TBX_AMI_TILE_PH * ty, TBX_AMI_TILE_PW, TBX_AMI_TILE_PH,
                 t == TBX_TILE_PAINTED ? c->painted_color : c->unpainted_color);
        }
    /* enemies in index order, then the player */
    for (int i = 0; i < s->n_enemies; i++) {
        const tbx_amidar_mover_t* m = &s->enemies[i];
        if (m->caught) continue;
        rect(out, ch, TBX_AMI_BOARD_OX + world_to_px(m->x) - 1, TBX_AMI_BOARD_OY + world_to_px(m->y) - 1, TBX_AMI_MOVER_W, TBX_AMI_MOVER_H, c->enemy_color);
    }
    rect(out, ch, TBX_AMI_BOARD_OX + world_to_px(s->player.x) - 1, TBX_AMI_BOARD_OY + world_to_px(s->player.y) - 1,
         TBX_AMI_MOVER_W, TBX_AMI_MOVER_H, c->player_color);
    /* HUD: score, lives, jumps, level */
    int sc = s->score; if (sc < 0) sc = 0; sc %= 100000;
    int div = 10000;
    for (int i = 0; i < 5; i++) { digit(out, ch, 20 + 8 * i, TBX_AMI_HUD_Y, (sc / div) % 10, c->player_color); div /= 10; }
    int lv = s->lives; if (lv < 0) lv = 0; if (lv > 9) lv = 9;
    digit(out, ch, 84, TBX_AMI_HUD_Y, lv, c->player_color);
    int jp = s->jumps; if (jp < 0) jp = 0; if (jp > 9) jp = 9;
    digit(out, ch, 108, TBX_AMI_HUD_Y, jp, c->player_color);
    int le = s->level; if (le < 0) le = 0;
    digit(out, ch, 132, TBX_AMI_HUD_Y, le % 10, c->player_color);
}
