/*
 * queue.c -- the bounded IQ-buffer queue between an SDR source thread and a DSP worker thread.
 * Same names and observable behaviour as the reference's src/queue.{h,c} (API src/queue.h:10-18):
 *   - queue_put copies the caller's samples into a free slot; a file source blocks while the queue is full,
 *     a live source overwrites the NEWEST filled slot and logs "<3>queue is full"   (queue.c:99-154)
 *   - take_buffer_for_processing detaches the OLDEST filled slot so the consumer can use it without the
 *     lock; complete_buffer_processing recycles it                                    (queue.c:168-213)
 *   - interrupt_waiting_the_data is the poison pill: pending slots are still delivered, then NULL
 *                                                                                     (queue.c:215-223)
 * Written from scratch around index rings (no linked nodes); slots are allocated once at create.
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sdrmodem_hip.h"

struct slot {
    sdrm_cf32 *samples;
    size_t len;
};

struct queue_t {
    struct slot *slots;
    uint16_t n_slots;
    uint32_t buffer_size;
    bool blocking;
    bool poisoned;
    /* FIFO of filled slot indices */
    uint16_t *filled;
    uint16_t filled_head, filled_count;
    /* stack of free slot indices */
    uint16_t *spare;
    uint16_t spare_count;
    int detached; /* slot being processed, -1 if none */
    pthread_mutex_t lock;
    pthread_cond_t changed;
};

int create_queue(uint32_t buffer_size, uint16_t queue_size, bool blocking, queue **out) {
    if (queue_size == 0) {
        fprintf(stderr, "<3>invalid queue size: %d\n", queue_size);
        return -1;
    }
    if (buffer_size == 0) {
        fprintf(stderr, "<3>invalid buffer size: %u\n", buffer_size);
        return -1;
    }
    struct queue_t *q = calloc(1, sizeof(*q));
    if (q == NULL) {
        return -ENOMEM;
    }
    q->slots = calloc(queue_size, sizeof(struct slot));
    q->filled = calloc(queue_size, sizeof(uint16_t));
    q->spare = calloc(queue_size, sizeof(uint16_t));
    if (q->slots == NULL || q->filled == NULL || q->spare == NULL) {
        free(q->slots);
        free(q->filled);
        free(q->spare);
        free(q);
        return -ENOMEM;
    }
    q->n_slots = queue_size;
    q->buffer_size = buffer_size;
    q->blocking = blocking;
    q->detached = -1;
    for (uint16_t i = 0; i < queue_size; i++) {
        q->slots[i].samples = malloc(sizeof(sdrm_cf32) * buffer_size);
        if (q->slots[i].samples == NULL) {
            for (uint16_t j = 0; j < i; j++) {
                free(q->slots[j].samples);
            }
            free(q->slots);
            free(q->filled);
            free(q->spare);
            free(q);
            return -ENOMEM;
        }
        /* hand slots out in creation order */
        q->spare[queue_size - 1 - i] = i;
    }
    q->spare_count = queue_size;
    pthread_mutex_init(&q->lock, NULL);
    pthread_cond_init(&q->changed, NULL);
    *out = q;
    return 0;
}

int queue_put(const sdrm_cf32 *buffer, size_t len, queue *q) {
    if (buffer == NULL || len == 0) {
        return -1;
    }
    if (len > q->buffer_size) {
        fprintf(stderr, "<3>requested buffer %zu is more than max: %d\n", len, q->buffer_size);
        return -1;
    }
    pthread_mutex_lock(&q->lock);
    if (q->blocking) {
        while (!q->poisoned && q->spare_count == 0) {
            pthread_cond_wait(&q->changed, &q->lock);
        }
        if (q->poisoned) {
            pthread_mutex_unlock(&q->lock);
            return -1;
        }
    }
    int target;
    if (q->spare_count > 0) {
        target = q->spare[--q->spare_count];
        q->filled[(q->filled_head + q->filled_count) % q->n_slots] = (uint16_t) target;
        q->filled_count++;
    } else {
        fprintf(stderr, "<3>queue is full\n");
        if (q->filled_count == 0) {
            /* every slot is detached for processing: nothing to overwrite, the buffer is lost
             * (the reference dereferences a NULL node here) */
            pthread_mutex_unlock(&q->lock);
            return 0;
        }
        target = q->filled[(q->filled_head + q->filled_count - 1) % q->n_slots];
    }
    memcpy(q->slots[target].samples, buffer, sizeof(sdrm_cf32) * len);
    q->slots[target].len = len;
    pthread_cond_broadcast(&q->changed);
    pthread_mutex_unlock(&q->lock);
    return 0;
}

void take_buffer_for_processing(sdrm_cf32 **buffer, size_t *len, queue *q) {
    pthread_mutex_lock(&q->lock);
    while (q->filled_count == 0 && !q->poisoned) {
        pthread_cond_wait(&q->changed, &q->lock);
    }
    if (q->filled_count == 0) {
        pthread_mutex_unlock(&q->lock);
        *buffer = NULL;
        return;
    }
    int idx = q->filled[q->filled_head];
    q->filled_head = (uint16_t) ((q->filled_head + 1) % q->n_slots);
    q->filled_count--;
    q->detached = idx;
    *buffer = q->slots[idx].samples;
    *len = q->slots[idx].len;
    pthread_mutex_unlock(&q->lock);
}

void complete_buffer_processing(queue *q) {
    pthread_mutex_lock(&q->lock);
    if (q->detached >= 0) {
        q->spare[q->spare_count++] = (uint16_t) q->detached;
        q->detached = -1;
    }
    pthread_cond_broadcast(&q->changed);
    pthread_mutex_unlock(&q->lock);
}

void interrupt_waiting_the_data(queue *q) {
    if (q == NULL) {
        return;
    }
    pthread_mutex_lock(&q->lock);
    q->poisoned = true;
    pthread_cond_broadcast(&q->changed);
    pthread_mutex_unlock(&q->lock);
}

void destroy_queue(queue *q) {
    if (q == NULL) {
        return;
    }
    for (uint16_t i = 0; i < q->n_slots; i++) {
        free(q->slots[i].samples);
    }
    free(q->slots);
    free(q->filled);
    free(q->spare);
    pthread_mutex_destroy(&q->lock);
    pthread_cond_destroy(&q->changed);
    free(q);
}
