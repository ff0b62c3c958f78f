/* oracle/hex.c -- CPU restatement of azalea/game/hex.py.  TEST INFRASTRUCTURE ONLY (see oracle.h). */
#include "oracle.h"

#include <string.h>

/* hex.py:146-149 */
void ohex_init(ohex_t *g, int n) {
    memset(g, 0, sizeof(*g));
    g->n = n;
    g->color = 1;
    g->winner = 0;
}

/* hex.py:151-159: ascending flat index + 1 of the empty cells; empty once there is a winner */
int ohex_legal_moves(const ohex_t *g, int32_t *out) {
    if (g->winner) return 0;
    int k = 0, cells = g->n * g->n;
    for (int i = 0; i < cells; ++i)
        if (g->board[i] == 0) out[k++] = i + 1;
    return k;
}

/* hex.py:161-170: 0 ongoing, 1 second player (O) won, 3 first player (X) won */
int ohex_result(const ohex_t *g) {
    if (g->winner) return g->winner == 2 ? 1 : 3;
    return 0;
}

/* hex.py:182-201: neighbour order (r-1,c) (r-1,c+1) (r,c-1) (r,c+1) (r+1,c-1) (r+1,c) */
static int neighbors(int tile, int n, int *out) {
    static const int di[6] = {-1, -1, 0, 0, 1, 1};
    static const int dj[6] = {0, 1, -1, 1, -1, 0};
    int ti = tile / n, tj = tile % n, k = 0;
    for (int d = 0; d < 6; ++d) {
        int ni = ti + di[d], nj = tj + dj[d];
        if (ni >= 0 && ni < n && nj >= 0 && nj < n) out[k++] = ni * n + nj;
    }
    return k;
}

/* hex.py:204-231: flood fill of the just-played tile's group; colour 1 tracks the row index
 * (wins top<->bottom), colour 2 the column index (wins left<->right). */
int ohex_check_win(const int32_t *board, int n, int tile) {
    int color = board[tile];
    if (!color) return 0;
    unsigned char seen[OHEX_MAXC];
    int stack[OHEX_MAXC * 6 + 1];
    int sp = 0, imin = 9999, imax = -9999;
    memset(seen, 0, sizeof(seen));
    stack[sp++] = tile;
    while (sp) {
        int t = stack[--sp];
        seen[t] = 1;
        int i = (color == 1) ? t / n : t % n;
        if (i < imin) imin = i;
        if (i > imax) imax = i;
        if (imin == 0 && imax == n - 1) return color;
        int nb[6];
        int k = neighbors(t, n, nb);
        for (int d = 0; d < k; ++d) {
            if (seen[nb[d]]) continue;
            if (board[nb[d]] == color && sp < OHEX_MAXC * 6) stack[sp++] = nb[d];
        }
    }
    return 0;
}

/* hex.py:172-179; returns 0 ok, -1 illegal (the reference asserts) */
int ohex_step(ohex_t *g, int move) {
    int tile = move - 1;
    if (tile < 0 || tile >= g->n * g->n || g->board[tile] != 0 || g->winner != 0) return -1;
    g->board[tile] = g->color;
    g->color = 3 - g->color;
    g->winner = ohex_check_win(g->board, g->n, tile);
    return 0;
}

/* hex.py:72-122: colour swap (b>0)*(3-b); mirror along the anti-diagonal
 * out[i][j] = in[n-1-j][n-1-i]; move (r,c) -> (n-1-c, n-1-r), list order and padding kept. */
void ohex_flip_board_moves(int n, const int32_t *board, const int32_t *moves, int K,
                           int32_t *fboard, int32_t *fmoves) {
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            int32_t b = board[(n - 1 - j) * n + (n - 1 - i)];
            fboard[i * n + j] = b > 0 ? 3 - b : 0;
        }
    for (int k = 0; k < K; ++k) {
        int32_t m = moves[k];
        if (m > 0) {
            int t = m - 1, r = t / n, c = t % n;
            fmoves[k] = (n - 1 - c) * n + (n - 1 - r) + 1;
        } else {
            fmoves[k] = 0;
        }
    }
}

/* azalea/fnv1a.py:6-20: 32-bit FNV-1a over the 4 little-endian bytes of each int32 */
uint32_t ofnv1a(const int32_t *seq, int len) {
    uint32_t h = 0x811c9dc5u;
    for (int i = 0; i < len; ++i) {
        uint32_t u = (uint32_t)seq[i];
        h = (h ^ (u & 0xff)) * 0x01000193u;
        h = (h ^ ((u >> 8) & 0xff)) * 0x01000193u;
        h = (h ^ ((u >> 16) & 0xff)) * 0x01000193u;
        h = (h ^ ((u >> 24) & 0xff)) * 0x01000193u;
    }
    return h;
}
