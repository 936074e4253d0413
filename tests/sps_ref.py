"""TEST-ONLY scalar restatement of the reference's SPS agent
(algorithms/v2x_sps.py:8-104) with its global-RNG calls replaced by injected
draws, so the device kernel can be compared decision for decision."""


class SpsRef:
    def __init__(self, prev_action, counter, rssi_threshold, inc_db=3, keep=0.8):
        self.prev_action = int(prev_action)             # v2x_sps.py:17
        self.reselection_counter = int(counter)         # v2x_sps.py:15
        self.RSSI_threshold = rssi_threshold            # v2x_sps.py:12
        self.inc_dB = inc_db                            # v2x_sps.py:18
        self.prob_resource_keep = keep                  # v2x_sps.py:22

    def choose_new_resource(self, selection_window, draw_choice):      # v2x_sps.py:24-74
        sB = []
        tmp_threshold = self.RSSI_threshold
        min_sA = len(selection_window) / 5
        sA = {}
        while len(sA) < min_sA:
            sA = {}
            for subframe in range(len(selection_window)):
                if self.prev_action == subframe:
                    continue
                if selection_window[subframe] < tmp_threshold:
                    sA[subframe] = selection_window[subframe]
            tmp_threshold += self.inc_dB
        sorted_sA = sorted(sA.items(), key=lambda x: x[1])
        min_len = min(min_sA, len(sA))
        for k, v in sorted_sA:
            sB.append(k)
            if len(sB) >= min_len:
                break
        return sB[draw_choice % len(sB)]                               # random.choice(sB)

    def step(self, selection_window, draw_counter, draw_keep, draw_choice):   # v2x_sps.py:76-104
        if self.reselection_counter != 0:
            action = self.prev_action
            self.reselection_counter -= 1
        else:
            self.reselection_counter = draw_counter                    # random.randint(5, 16)
            if draw_keep < self.prob_resource_keep:                    # random.random()
                action = self.prev_action
            else:
                action = self.choose_new_resource(selection_window, draw_choice)
                self.prev_action = action
        return action
